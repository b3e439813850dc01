// Similarity-and-reduce on the matrix cores (simred.h): sim = alpha * A . B^T is computed 128 x 128 tile by tile and reduced on the spot.
//
// Why not the tile GEMM with a reducing epilogue (gemm.hip, EPI_NNSTAT / EPI_SIMSTAT; rounds 3-4)?  With K = 128 .. 256 a tile kernel
// spends its life in prologues and epilogues: both operands fetched from HBM / L2 per tile and split into f16 planes on the fly
// (the A panel 40 .. 128 times), four to eight k-tiles, then the epilogue -- 0.21 .. 0.31 of the matrix pipe, 4.5 x over-fetch.  Here:
//   * both operands are split ONCE by a packing pass into ready-to-use MFMA fragments ([32 rows][16 k] blocks, 16 bytes per lane and
//     piece): no vector work on either operand inside the loop;
//   * a workgroup OWNS 128 rows of A for its whole life: a wave holds the fragments of its 32 (K = 256) or 64 (K <= 128) rows in
//     registers (128 VGPRs at most), fetched once;
//   * the B fragments of the column tiles STREAM through a four-slot LDS ring (16 KiB = two k-steps of a 128-column tile per stage, one
//     barrier per stage), brought in by LDS-DMA (`global_load_lds_dwordx4`: no registers, no VALU) THREE stages ahead: the pipeline never
//     drains between tiles and a request has three stages of matrix work to come back from L2 / the Infinity Cache (the first version
//     went through registers one stage ahead and ran at the latency of a load per stage: 0.29 of the matrix pipe);
//   * accumulators hold sim^T (B fragment = MFMA A operand): a lane owns one row i of sim per row fragment and 16 columns per
//     accumulator, so everything per ROW is lane-local and is carried in registers across all column tiles (no row partials, no merge
//     for the rows of a chunk);
//   * per COLUMN the whole tile is parked in LDS column-major, every column is reduced by four adjacent lanes (eight conflict-free
//     16-byte reads each) and folded with two DPP exchanges; one partial per column and 128-row block goes to memory (8 .. 12 bytes
//     per column and row block instead of 4 bytes per element).
// Measured and dropped: the two column halves half a tile apart (lab switch, bit 2 of IMCUI_SR_DBG), a fifth ring
// slot with the next stage's fragments read one barrier early.
// Product order per accumulator (b_hi a_lo, b_lo a_hi, b_hi a_hi; k ascending) is that of gemm_split_kernel, so a similarity has
// the bits the tile GEMM gave it.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "simred.h"

#define SR_PLD 132      // floats per parked COLUMN of 128 rows (+ 4: the 16-byte reads of four adjacent columns x four row groups hit 64 different banks)
#define SR_SLOT_U4 1024  // uint4 per ring slot: 4 column fragments x 2 k-steps x 2 pieces x 64 lanes
#define SR_NSLOT 4        // ring slots: the stage being multiplied + three in flight
#define SR_MAXLIST 2048  // tiles per chunk when tiles are selected by flags

__device__ __forceinline__ float sr_lg_score(float s, float rm, float rl, float cm, float cl, float l0, float l1) {
    return (((s - rm) - rl) + ((s - cm) - cl)) + (l0 + l1);  // the reference's association (lightglue_assign.h)
}

// ------------------------------------------------------------------ packing
// grid (fragments, 1, batch), 4 waves: wave w packs k-steps w, w + 4, ...; lane (lo, hi) owns row 32 f + lo, k = 16 ks + 8 hi .. + 7
template <bool F32>
__global__ __launch_bounds__(256) void simred_pack_kernel(const float* __restrict__ X, long ldr, long ldk, long xbs, int rows, const int* __restrict__ cnt,
                                                          int cnt_stride, int KS, uint4* __restrict__ out, long obs) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lo = lane & 31, hi = lane >> 5;
    const int f = blockIdx.x, b = blockIdx.z;
    const int nrow = cnt ? min(cnt[(size_t)b * cnt_stride], rows) : rows;
    const int row = f * 32 + lo;
    const float* src = X + (size_t)b * xbs + (size_t)row * ldr;
    uint4* dst = out + (size_t)b * obs + (size_t)f * KS * 128 + lane;
    for (int ks = wid; ks < KS; ks += 4) {
        float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
        if (row < nrow) {
            const int k0 = 16 * ks + 8 * hi;
            if (ldk == 1 && (ldr & 3) == 0) {
                va = *reinterpret_cast<const float4*>(src + k0);
                vb = *reinterpret_cast<const float4*>(src + k0 + 4);
            } else {
                va = make_float4(src[(size_t)k0 * ldk], src[(size_t)(k0 + 1) * ldk], src[(size_t)(k0 + 2) * ldk], src[(size_t)(k0 + 3) * ldk]);
                vb = make_float4(src[(size_t)(k0 + 4) * ldk], src[(size_t)(k0 + 5) * ldk], src[(size_t)(k0 + 6) * ldk], src[(size_t)(k0 + 7) * ldk]);
            }
        }
        uint4 p0, p1;
        if constexpr (F32) {
            p0 = __builtin_bit_cast(uint4, va);
            p1 = __builtin_bit_cast(uint4, vb);
        } else {
            split8(va, vb, p0, p1);  // the same split the tile GEMM applies while staging (hi toward zero, lo nearest)
        }
        dst[(size_t)ks * 128] = p0;
        dst[(size_t)ks * 128 + 64] = p1;
    }
}

void simred_pack(imcui_hip_s* h, const float* X, long ldr, long ldk, long xbs, int rows, int K, int batch, const int* cnt, int cnt_stride, uint4* out,
                 hipStream_t stream) {
    const int nfrag = (rows + SR_TILE - 1) / SR_TILE * (SR_TILE / 32);
    const long obs = (long)simred_packed_uint4(rows, K);
    const dim3 grid(nfrag, 1, batch);
    if (h->precision == 1)
        hipLaunchKernelGGL(simred_pack_kernel<false>, grid, dim3(256), 0, stream, X, ldr, ldk, xbs, rows, cnt, cnt_stride, K / 16, out, obs);
    else
        hipLaunchKernelGGL(simred_pack_kernel<true>, grid, dim3(256), 0, stream, X, ldr, ldk, xbs, rows, cnt, cnt_stride, K / 16, out, obs);
}

// ------------------------------------------------------------------ the kernel
// KS = K / 16.  512 threads = 8 waves, one workgroup per CU.  WN = column groups of waves:
//   WN = 2: wave (wm < 4, wn < 2) owns rows 32 wm .. + 31 x columns 64 wn .. + 63 of a tile (1 x 2 accumulator fragments; every width);
//   WN = 4 (K <= 128: the A fragments of 64 rows are 64 .. 128 VGPRs): wave (wm < 2, wn < 4) owns rows 64 wm .. + 63 x columns 32 wn ..
//           + 31 (2 x 1 fragments): a streamed fragment is read from LDS by TWO waves instead of four -- half the LDS traffic per matrix
//           instruction, which is what the loop waits for.
template <int KS, int WN, int MODE, bool F32>
__global__ __launch_bounds__(512, 2) void simred_kernel(SimRedP p) {
    constexpr int NW = 8, NWM = NW / WN, RF = 4 / NWM, NF = 4 / WN, KT = KS / 2, PPW = 16 / NW;
    constexpr bool ISNN = MODE == SR_NN || MODE == SR_NN1;  // SR_NN1: no second best (find_nn without a ratio test)
    constexpr bool PREMASK = ISNN || MODE == SR_LSE;       // columns past the matrix are set to -inf in the accumulators, before the epilogue
    static_assert(WN == 2 || WN == 4, "column groups of waves");
    static_assert(KT >= 2, "at least two stages per tile");
    extern __shared__ uint4 sr_smem[];
    uint4* ring = sr_smem;                                                     // SR_NSLOT slots
    float* park = reinterpret_cast<float*>(sr_smem + SR_NSLOT * SR_SLOT_U4);   // [128 columns][SR_PLD]
    float* ctab = park + SR_TILE * SR_PLD;                                      // [3][128] column constants of the tile (pass 2)
    unsigned short* tlist = reinterpret_cast<unsigned short*>(ctab + 3 * SR_TILE);
    __shared__ int s_ntl;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int wm = wid % NWM, wn = wid / NWM;
    const int nrb = (p.M + SR_TILE - 1) / SR_TILE, nct_s = (p.N + SR_TILE - 1) / SR_TILE;
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int rb = t % nrb;
    t /= nrb;
    const int chunk = t % p.nchunk, b = t / p.nchunk;
    const int Mb = p.mcnt ? min(p.mcnt[(size_t)b * p.cnt_stride], p.M) : p.M;
    const int Nb = p.ncnt ? min(p.ncnt[(size_t)b * p.cnt_stride], p.N) : p.N;
    if (rb * SR_TILE >= Mb) return;
    const int rowsv = min(SR_TILE, Mb - rb * SR_TILE);  // live rows of the block
    const int tpc = (nct_s + p.nchunk - 1) / p.nchunk;
    const int ct0 = chunk * tpc, ct1 = min(ct0 + tpc, (Nb + SR_TILE - 1) / SR_TILE);

    // ---- the tiles of this workgroup
    int ntl;
    const bool listed = (MODE == SR_DSBEST) && p.flags != nullptr;
    if (listed) {
        if (wid == 0) {
            const unsigned char* fl = p.flags + ((size_t)b * nrb + rb) * nct_s;
            int n = 0;
            for (int c0 = ct0; c0 < ct1; c0 += 64) {
                const int c = c0 + lane;
                const bool f = c < ct1 && fl[c] != 0;
                const unsigned long long bal = __ballot(f);
                if (f) tlist[n + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)c;
                n += __popcll(bal);
            }
            if (lane == 0) s_ntl = n;
        }
        __syncthreads();
        ntl = s_ntl;
    } else {
        ntl = max(ct1 - ct0, 0);
    }
    auto tile_at = [&](int i) -> int { return listed ? (int)tlist[i] : ct0 + i; };

    // ---- A fragments of the wave's 32 RF rows: registers, for the life of the workgroup
    uint4 a0[RF][KS], a1[RF][KS];
#pragma unroll
    for (int f = 0; f < RF; ++f) {
        const uint4* ap = p.Ap + (size_t)b * p.ap_bs + ((size_t)(rb * 4 + wm * RF + f) * KS) * 128 + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            a0[f][ks] = ap[ks * 128];
            a1[f][ks] = ap[ks * 128 + 64];
        }
    }
    // ---- B stages: piece pc = wid * PPW + u of a stage = fragment pc >> 2, (k-step, piece) pc & 3.  Everything but the lane offset is
    // wave-uniform and is kept scalar (readfirstlane tells the compiler so): one 32-bit lane offset against a scalar base per load,
    // instead of a 64-bit vector address per (piece, stage of a tile) hoisted out of the loop
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);
    const uint4* bpb = p.Bp + (size_t)b * p.bp_bs + ((size_t)((wid_s * PPW) >> 2) * KS) * 128 + ((wid_s * PPW) & 3) * 64;
    // LDS-DMA: one instruction moves 1 KiB (16 bytes per lane) from the lanes' global addresses to M0 + 16 lane; issued from inline asm
    // (M0 saved / restored around it) because a DMA the compiler can see makes it drain vmcnt to 0 in front of every LDS read -- the
    // three-stage lead is the point.  The waits are placed by hand below (sr_wait_dma).
    typedef __attribute__((address_space(3))) void* lptr_t;
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)ring) + (unsigned)(wid_s * PPW) * 1024u;
    auto dma_stage = [&](int ct, int kslice, int slot) __attribute__((always_inline)) {
        const uint4* src = bpb + ((size_t)ct * 4 * KS + kslice * 2) * 128 + lane;  // (scalar base + lane)
        const unsigned dst = ring_lds + (unsigned)slot * (SR_SLOT_U4 * 16u);
        unsigned keep;
        if constexpr (PPW == 2)
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                         "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(src), "v"(src + 64), "s"(dst), "s"(dst + 1024u)
                         : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                         "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                         "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                         "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(src), "v"(src + 64), "v"(src + 128), "v"(src + 192), "s"(dst), "s"(dst + 1024u), "s"(dst + 2048u), "s"(dst + 3072u)
                         : "memory");
    };
    // wait until at most `stages` of this wave's DMA stages are outstanding (requests retire in order; the stores of an epilogue may sit
    // between them -- they only make the wait longer, never shorter than needed)
    auto wait_dma = [&](int stages) __attribute__((always_inline)) {
        if (stages >= 2)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else if (stages == 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    constexpr int LKT = (KT == 2) ? 1 : (KT == 4) ? 2 : 3;  // log2(KT)
    static_assert((1 << LKT) == KT, "KS is 4, 8 or 16");

    f32x16 acc[RF][NF];
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
#pragma unroll
    for (int f = 0; f < RF; ++f)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[f][n] = zero16;

    // ---- per-row running state (lane = rows (wm RF + f) * 32 + lo, their columns 4 hi .. of every 8)
    const int rowl0 = wm * RF * 32 + lo, rowg0 = rb * SR_TILE + rowl0;  // fragment f: + 32 f
    float q0[RF], q1[RF];  // NN: best, second; LSE: max, sum; BEST: best value
    int qi[RF];
    float rc0[RF], rc1[RF], rc2[RF];  // pass 2: row constants
#pragma unroll
    for (int f = 0; f < RF; ++f) {
        q0[f] = (MODE == SR_DSBEST) ? -1.0f : -INFINITY;
        q1[f] = ISNN ? -INFINITY : 0.0f;
        qi[f] = 0x7fffffff;
        rc0[f] = rc1[f] = rc2[f] = 0.0f;
        if constexpr (MODE == SR_DSBEST || MODE == SR_LGBEST) {
            const int rg = min(rowg0 + 32 * f, Mb - 1);
            const size_t ro = (size_t)b * p.r_pitch + rg;
            rc0[f] = p.rmax[ro];
            rc1[f] = (MODE == SR_DSBEST) ? __builtin_amdgcn_rcpf(p.rsum[ro]) : p.rsum[ro];
            if (MODE == SR_LGBEST) rc2[f] = p.l0[(size_t)b * p.l0_bs + rg];
        }
    }

    const int S = ntl * KT;
    // stages 0 .. 2 on their way; stage 0 landed (everywhere: the barrier) before the loop starts
    for (int g = 0; g < 3 && g < S; ++g) dma_stage(tile_at(g >> LKT), g & (KT - 1), g);
    wait_dma(S >= 3 ? 2 : S >= 2 ? 1 : 0);
    __syncthreads();

    // ---- lab switch (IMCUI_SR_DBG bit 2; OFF by default, a measured negative result): the two column halves HALF A TILE APART.  A tile is KT
    // matrix intervals + two reducing intervals (per row + parking | per column), each closed by a workgroup barrier.  In step, the two
    // waves of a SIMD (same wm, wn = 0 / 1) both multiply, then both reduce, and the matrix pipe idles through every epilogue.  Nothing in
    // the loop couples the halves except the barriers (a wave's DMA pieces are the fragments of its own half, the parked tile is scanned
    // by the half that parked it, the column constants are per half), so the wn = 1 waves can simply enter the loop `lag` barriers later
    // and the wn = 0 waves pass `lag` barriers after it: while one wave of a SIMD reduces, the other multiplies.  Measured (MI355X, round
    // 5): K = 128 nearest neighbours 1.68 -> 1.79 ms, LightGlue's assignment pass 0.56 -> 0.67 ms, the K = 256 statistics pass 7.21 ->
    // 7.00 ms.  The loop is not bound by issue slots but by the LDS pipe (every B fragment is read by the four row waves: 64 KB per stage
    // = 512 of a stage's 768 matrix cycles, + 128 KB per parked tile) and by the barriers; apart, every barrier waits for the slower of a
    // matrix and a reducing interval and the two kinds of LDS traffic collide.
    const int lag = (WN == 2 && (p.dbg & 4)) ? (KT + 2) / 2 : 0;
    if (__builtin_amdgcn_readfirstlane(wn) == 1)
        for (int g = 0; g < lag; ++g) __builtin_amdgcn_s_barrier();

    for (int ti = 0; ti < ntl; ++ti) {
        const int ct = tile_at(ti);
        if constexpr (MODE == SR_DSBEST || MODE == SR_LGBEST) {
            // column constants of the tile (read in the epilogue, >= 1 barrier from here; the previous tile's readers are past its last barrier)
            // (each column half loads -- and later reads -- its own 64 entries: the halves are not at the same tile, see `lag`)
            if (wm * 64 + lane < NF * 32) {  // (the first threads of every column group: its own NF * 32 entries)
                const int jl = wn * (NF * 32) + wm * 64 + lane;
                const int j = min(ct * SR_TILE + jl, Nb - 1);
                const size_t co = (size_t)b * p.c_pitch + j;
                ctab[jl] = p.cmax[co];
                ctab[SR_TILE + jl] = (MODE == SR_DSBEST) ? __builtin_amdgcn_rcpf(p.csum[co]) : p.csum[co];
                if (MODE == SR_LGBEST) ctab[2 * SR_TILE + jl] = p.l1[(size_t)b * p.l1_bs + j];
            }
        }
#pragma unroll
        for (int kslice = 0; kslice < KT; ++kslice) {
            const int s = ti * KT + kslice;
            const int slot = (KT >= SR_NSLOT) ? (kslice & (SR_NSLOT - 1)) : (s & (SR_NSLOT - 1));
            // stage s + 3 into the slot stage s - 1 was multiplied from (every wave is past the barrier that ended it)
            if (s + 3 < S) dma_stage(tile_at((s + 3) >> LKT), (kslice + 3) & (KT - 1), (slot + 3) & (SR_NSLOT - 1));
#pragma unroll
            for (int ksl = 0; ksl < 2; ++ksl) {
                const int ks = kslice * 2 + ksl;
                uint4 b0[NF], b1[NF];
#pragma unroll
                for (int n = 0; n < NF; ++n) {
                    const int nf = wn * NF + n;
                    b0[n] = ring[slot * SR_SLOT_U4 + (nf * 4 + ksl * 2) * 64 + lane];
                    b1[n] = ring[slot * SR_SLOT_U4 + (nf * 4 + ksl * 2 + 1) * 64 + lane];
                }
                if constexpr (!F32) {
                    // pieces = f16 hi / lo planes; per accumulator: b_hi a_lo, b_lo a_hi, b_hi a_hi (gemm_split_kernel's order), the NF
                    // accumulators interleaved so consecutive matrix instructions are independent
                    // (the first product of a tile takes a constant-zero C operand: the accumulators are never cleared by vector moves)
#pragma unroll
                    for (int f = 0; f < RF; ++f)
#pragma unroll
                        for (int n = 0; n < NF; ++n) acc[f][n] = mfma16(b0[n], a1[f][ks], ks == 0 ? zero16 : acc[f][n]);
#pragma unroll
                    for (int f = 0; f < RF; ++f)
#pragma unroll
                        for (int n = 0; n < NF; ++n) acc[f][n] = mfma16(b1[n], a0[f][ks], acc[f][n]);
#pragma unroll
                    for (int f = 0; f < RF; ++f)
#pragma unroll
                        for (int n = 0; n < NF; ++n) acc[f][n] = mfma16(b0[n], a0[f][ks], acc[f][n]);
                } else {
                    // pieces = the lane's two k-quads; step j pairs k = 16 ks + j (lanes hi = 0) with k = 16 ks + 8 + j (hi = 1)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int f = 0; f < RF; ++f) {
                            const float4 fa = __builtin_bit_cast(float4, j < 4 ? a0[f][ks] : a1[f][ks]);
                            const float av4[4] = {fa.x, fa.y, fa.z, fa.w};
#pragma unroll
                            for (int n = 0; n < NF; ++n) {
                                const float4 fb = __builtin_bit_cast(float4, j < 4 ? b0[n] : b1[n]);
                                const float bv4[4] = {fb.x, fb.y, fb.z, fb.w};
                                acc[f][n] = mfma32(bv4[j & 3], av4[j & 3], (ks == 0 && j == 0) ? zero16 : acc[f][n]);
                            }
                        }
                }
            }
            if (kslice + 1 < KT) {
                // stage s + 1 of THIS wave has landed (two younger stages may still be in flight); the barrier makes it everybody's
                wait_dma(s + 3 < S ? 2 : s + 2 < S ? 1 : 0);
                __syncthreads();
            } else if (lag) {
                __builtin_amdgcn_s_barrier();  // (closes the last matrix interval: the reducing intervals line up with the other half's matrix intervals)
            }
        }

        // ================================================================ epilogue of tile ct
        // The thread index is laundered through an empty asm so that the epilogue's address arithmetic is recomputed per tile (a few
        // dozen instructions) instead of being hoisted out of the tile loop and carried -- 60 VGPRs of it -- across the MFMA loop.
        int etid = tid;
        asm volatile("" : "+v"(etid));
        const int elo = etid & 31, ehi = (etid >> 5) & 1, erowl0 = (((etid >> 6) % NWM) * RF) * 32 + elo;  // row of fragment f: + 32 f
        // (1) per row, from the accumulators: lane = rows rowl0 + 32 f, columns jb + 32 n + 8 q + e; the value to park replaces the accumulator
        const int jb = ct * SR_TILE + wn * (NF * 32) + 4 * ehi;
        // (ONE instance of this code: a variant without the column mask for the tiles that lie inside the matrix was tried twice -- as a
        // second copy of the masking part and as a second copy of the whole epilogue -- and both made hipcc spill: the accumulators are
        // rewritten in place and two variants of that meeting in one control-flow join keep both register sets alive)
        const int jlim = Nb - jb;  // column c of the lane's list is live when c < jlim
        if constexpr (PREMASK) {
            // only the last column tile of a matrix has dead columns: they are set to -inf IN PLACE, in a block of its own, so the common
            // tile pays neither the compare nor the select per element (a second copy of the epilogue for it made hipcc spill: see above)
            if (__builtin_amdgcn_readfirstlane(Nb - ct * SR_TILE) < SR_TILE) {
#pragma unroll
                for (int f = 0; f < RF; ++f)
#pragma unroll
                    for (int n = 0; n < NF; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[f][n][r] = (n * 32 + 8 * (r >> 2) + (r & 3) < jlim) ? acc[f][n][r] : -INFINITY;
            }
        }
        if (!(p.dbg & 2))
#pragma unroll
        for (int f = 0; f < RF; ++f) {
        float tmax = -INFINITY;
        const float q0old = q0[f];
        int tc = -1;  // column of the row's new best inside this tile (relative to jb), -1: the best did not move
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float cm4[4] = {0.f, 0.f, 0.f, 0.f}, cs4[4] = {0.f, 0.f, 0.f, 0.f}, cl4[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (MODE == SR_DSBEST || MODE == SR_LGBEST) {
                    const int jl = wn * (NF * 32) + 4 * ehi + n * 32 + 8 * q;
                    const float4 t0 = *reinterpret_cast<const float4*>(ctab + jl), t1 = *reinterpret_cast<const float4*>(ctab + SR_TILE + jl);
                    cm4[0] = t0.x, cm4[1] = t0.y, cm4[2] = t0.z, cm4[3] = t0.w;
                    cs4[0] = t1.x, cs4[1] = t1.y, cs4[2] = t1.z, cs4[3] = t1.w;
                    if (MODE == SR_LGBEST) {
                        const float4 t2 = *reinterpret_cast<const float4*>(ctab + 2 * SR_TILE + jl);
                        cl4[0] = t2.x, cl4[1] = t2.y, cl4[2] = t2.z, cl4[3] = t2.w;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q + e;
                    const bool valid = n * 32 + 8 * q + e < jlim;
                    const float x = PREMASK ? 0.0f : acc[f][n][r] * p.alpha;
                    (void)valid, (void)x;
                    if constexpr (ISNN) {
                        // (alpha is 1 for the nearest-neighbour modes: simred_launch refuses anything else; dead columns are -inf already)
                        if constexpr (MODE == SR_NN) {
                            q1[f] = __builtin_amdgcn_fmed3f(q0[f], q1[f], acc[f][n][r]);  // second best of {best, second, x} (second <= best)
                            q0[f] = fmaxf(q0[f], acc[f][n][r]);
                        }
                    } else if constexpr (MODE == SR_LSE) {
                        const float xv = acc[f][n][r] * p.alpha;  // (alpha > 0: a dead column stays -inf)
                        tmax = fmaxf(tmax, xv);
                        acc[f][n][r] = xv;
                    } else if constexpr (MODE == SR_DSBEST) {
                        // conf = softmax over i (column statistics) * softmax over j (row statistics), evaluated ONCE per element
                        const float v = valid ? (__expf(x - cm4[e]) * cs4[e]) * (__expf(x - rc0[f]) * rc1[f]) : -1.0f;
                        tmax = fmaxf(tmax, v);
                        acc[f][n][r] = v;
                    } else {
                        const float v = valid ? sr_lg_score(x, rc0[f], rc1[f], cm4[e], cs4[e], rc2[f], cl4[e]) : -INFINITY;
                        tmax = fmaxf(tmax, v);
                        acc[f][n][r] = v;
                    }
                }
                // one group of four columns at a time: left alone, hipcc's scheduler starts all 64 elements at once (every temporary of
                // every element live together) and the kernel spills
                __builtin_amdgcn_sched_barrier(0);
            }
        if constexpr (ISNN) {
            // (q0 is the row's best including this tile: if it moved, its first column in the tile is found by equality, walking downwards)
            float tm = acc[f][0][0];
#pragma unroll
            for (int n = 0; n < NF; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) tm = fmaxf(tm, acc[f][n][r]);
#pragma unroll
            for (int n = NF - 1; n >= 0; --n)
#pragma unroll
                for (int r = 15; r >= 0; --r) tc = (acc[f][n][r] == tm) ? n * 32 + 8 * (r >> 2) + (r & 3) : tc;
            qi[f] = tm > q0old ? jb + tc : qi[f];
            if (MODE == SR_NN1) q0[f] = fmaxf(q0old, tm);
        }
        if constexpr (MODE == SR_DSBEST || MODE == SR_LGBEST) {
            // the row's best of this tile is a maximum tree over the finished values; its FIRST column is found by equality, walking the
            // columns downwards (the last hit is the lowest column) -- no chain through the running best while the values are formed
#pragma unroll
            for (int n = NF - 1; n >= 0; --n)
#pragma unroll
                for (int r = 15; r >= 0; --r) tc = (acc[f][n][r] == tmax) ? n * 32 + 8 * (r >> 2) + (r & 3) : tc;
            const bool up = tmax > q0[f];  // tiles ascend: a later tile wins only with a larger value
            qi[f] = up ? jb + tc : qi[f];
            q0[f] = up ? tmax : q0[f];
        }
        if constexpr (MODE == SR_LSE) {
            // online (max, sum): the reference moves once per tile
            const float mn = fmaxf(q0[f], tmax);
            const float mref = (mn == -INFINITY) ? 0.0f : mn;
            float sum = q1[f] * __expf(q0[f] - mref);  // q0 = -inf: 0
#pragma unroll
            for (int n = 0; n < NF; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += __expf(acc[f][n][r] - mref);
            q1[f] = sum;
            q0[f] = mn;
        }
        }  // (row fragment f)
        // (2) per column: the whole tile is parked COLUMN-major ([column][row], 132-float rows) and every column is reduced by FOUR adjacent
        // lanes: thread (column jl = t >> 2, part = t & 3) reads rows 64 hh + 16 part + 4 k .. + 3 (hh < 2, k < 4) as eight conflict-free
        // 16-byte reads -- 32 values, no loop, no dependent LDS latency -- reduces them in registers, and the four parts are folded with two
        // DPP quad exchanges; lane part 0 stores.  One parking round and two barriers per tile.  (The first version parked row-major, 64
        // columns at a time, scanned with (column, 16-row range) threads in a loop and folded the eight ranges through LDS by 64 threads
        // while the other seven waves waited: 42 % of the LoFTR statistics launch, measured with IMCUI_SR_DBG.)
        if (!(p.dbg & 1)) {
#pragma unroll
            for (int f = 0; f < RF; ++f)
#pragma unroll
                for (int n = 0; n < NF; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        park[(wn * (NF * 32) + n * 32 + 8 * (r >> 2) + (r & 3) + 4 * ehi) * SR_PLD + erowl0 + 32 * f] = acc[f][n][r];
            __syncthreads();
            {
                const int jl = etid >> 2, part = etid & 3;
                const float4* col = reinterpret_cast<const float4*>(park + jl * SR_PLD + 16 * part);
                float v[32];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float4 t4 = col[16 * hh + k];  // rows 64 hh + 16 part + 4 k .. + 3
                        v[16 * hh + 4 * k + 0] = t4.x, v[16 * hh + 4 * k + 1] = t4.y, v[16 * hh + 4 * k + 2] = t4.z, v[16 * hh + 4 * k + 3] = t4.w;
                    }
                constexpr float NEUTRAL = (MODE == SR_DSBEST) ? -1.0f : -INFINITY;
                if (rowsv < SR_TILE) {  // (the last row block of a matrix: rows past its end are neutral)
#pragma unroll
                    for (int u = 0; u < 32; ++u) v[u] = (64 * (u >> 4) + 16 * part + (u & 15) < rowsv) ? v[u] : NEUTRAL;
                }
                // element u of the list is row 64 (u >> 4) + 16 part + (u & 15): ascending in u
                float tm = v[0];
#pragma unroll
                for (int u = 1; u < 32; ++u) tm = fmaxf(tm, v[u]);
                const int j = ct * SR_TILE + jl;
                const size_t o = ((size_t)b * nrb + rb) * p.c_pitch + j;
                const int rbase = rb * SR_TILE + 16 * part;
                if constexpr (ISNN || MODE == SR_LGBEST) {
                    // first row attaining the maximum: equality, walking downwards (the last hit is the lowest row)
                    int ti = 0x7fffffff - rbase;
#pragma unroll
                    for (int u = 31; u >= 0; --u) ti = (v[u] == tm) ? 64 * (u >> 4) + (u & 15) : ti;
                    ti += rbase;
                    float b2 = -INFINITY;
                    if constexpr (MODE == SR_NN) {
                        // second best = the maximum of the list with ONE occurrence of the best removed: the running pair over the values
                        float b1 = -INFINITY;
#pragma unroll
                        for (int u = 0; u < 32; ++u) {
                            b2 = __builtin_amdgcn_fmed3f(b1, b2, v[u]);
                            b1 = fmaxf(b1, v[u]);
                        }
                    }
                    // fold the four parts (interleaved row sets): larger value, lowest row on equal values
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const int ctrl = st == 0 ? 0xB1 : 0x4E;  // quad_perm [1,0,3,2] / [2,3,0,1]: lane ^ 1, lane ^ 2
                        const float om = __builtin_bit_cast(float, st == 0 ? __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, tm), 0xB1, 0xF, 0xF, true)
                                                                          : __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, tm), 0x4E, 0xF, 0xF, true));
                        const int oi = st == 0 ? __builtin_amdgcn_mov_dpp(ti, 0xB1, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(ti, 0x4E, 0xF, 0xF, true);
                        const float o2 = __builtin_bit_cast(float, st == 0 ? __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, b2), 0xB1, 0xF, 0xF, true)
                                                                          : __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, b2), 0x4E, 0xF, 0xF, true));
                        (void)ctrl;
                        const bool up = om > tm || (om == tm && oi < ti);
                        if (MODE == SR_NN) b2 = up ? fmaxf(tm, o2) : fmaxf(b2, om);
                        (void)o2;
                        ti = up ? oi : ti;
                        tm = up ? om : tm;
                    }
                    if (part == 0 && j < Nb) {
                        p.c0[o] = tm;
                        if (MODE == SR_NN) p.c1[o] = b2;
                        p.ci[o] = ti;
                    }
                } else if constexpr (MODE == SR_LSE) {
                    const float mref = (tm == -INFINITY) ? 0.0f : tm;
                    float sum = 0.0f;
#pragma unroll
                    for (int u = 0; u < 32; ++u) sum += __expf(v[u] - mref);
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const float om = __builtin_bit_cast(float, st == 0 ? __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, tm), 0xB1, 0xF, 0xF, true)
                                                                          : __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, tm), 0x4E, 0xF, 0xF, true));
                        const float os = __builtin_bit_cast(float, st == 0 ? __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, sum), 0xB1, 0xF, 0xF, true)
                                                                          : __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, sum), 0x4E, 0xF, 0xF, true));
                        const float mm = fmaxf(tm, om);
                        const float mr = (mm == -INFINITY) ? 0.0f : mm;
                        sum = sum * __expf(tm - mr) + os * __expf(om - mr);  // (both lanes of a pair evaluate the same expression on swapped operands:
                        tm = mm;                                             //  the sum is commutative in exact arithmetic only -- lane `part 0` decides)
                    }
                    if (part == 0 && j < Nb) {
                        p.c0[o] = tm;
                        p.c1[o] = sum;
                    }
                } else {
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const float om = __builtin_bit_cast(float, st == 0 ? __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, tm), 0xB1, 0xF, 0xF, true)
                                                                          : __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, tm), 0x4E, 0xF, 0xF, true));
                        tm = fmaxf(tm, om);
                    }
                    if (part == 0 && j < Nb) p.c0[o] = tm;
                }
            }
        }
        {
            const int s = ti * KT + KT - 1;
            wait_dma(s + 3 < S ? 2 : s + 2 < S ? 1 : 0);
        }
        __syncthreads();
    }

    if (__builtin_amdgcn_readfirstlane(wn) == 0)
        for (int g = 0; g < lag; ++g) __builtin_amdgcn_s_barrier();
    __syncthreads();  // (both halves out of the loop: the parked tiles are dead, `park` carries the row exchange below)

    // ================================================================ rows: fold the two half-waves (columns 4 hi), then the WN column groups
    // (one partial against another: interleaved / ascending column sets -- lowest index on equal values)
    auto fold = [&](float& a0_, float& a1_, int& ai_, float o0, float o1, int oi) __attribute__((always_inline)) {
        if constexpr (ISNN) {
            const bool up = o0 > a0_ || (o0 == a0_ && oi < ai_);
            a1_ = up ? fmaxf(a0_, o1) : fmaxf(a1_, o0);
            ai_ = up ? oi : ai_;
            a0_ = up ? o0 : a0_;
        } else if constexpr (MODE == SR_LSE) {
            const float m = fmaxf(a0_, o0);
            const float mref = (m == -INFINITY) ? 0.0f : m;
            a1_ = a1_ * __expf(a0_ - mref) + o1 * __expf(o0 - mref);
            a0_ = m;
        } else {
            const bool up = o0 > a0_ || (o0 == a0_ && oi < ai_);
            ai_ = up ? oi : ai_;
            a0_ = up ? o0 : a0_;
        }
    };
#pragma unroll
    for (int f = 0; f < RF; ++f) fold(q0[f], q1[f], qi[f], __shfl_xor(q0[f], 32, 64), __shfl_xor(q1[f], 32, 64), __shfl_xor(qi[f], 32, 64));
    {
        float* xr = park;  // [WN - 1][128 rows][4]  (every scan of the loop is behind its last barrier)
        if (wn > 0 && hi == 0) {
#pragma unroll
            for (int f = 0; f < RF; ++f) {
                const int o = ((wn - 1) * SR_TILE + rowl0 + 32 * f) * 4;
                xr[o + 0] = q0[f];
                xr[o + 1] = q1[f];
                reinterpret_cast<int*>(xr)[o + 2] = qi[f];
            }
        }
        __syncthreads();
        if (wn == 0 && hi == 0) {
#pragma unroll
            for (int f = 0; f < RF; ++f)
#pragma unroll
                for (int g = 1; g < WN; ++g) {  // ascending column groups
                    const int o = ((g - 1) * SR_TILE + rowl0 + 32 * f) * 4;
                    fold(q0[f], q1[f], qi[f], xr[o + 0], xr[o + 1], reinterpret_cast<int*>(xr)[o + 2]);
                }
        }
    }
    if (wn == 0 && hi == 0) {
#pragma unroll
        for (int f = 0; f < RF; ++f)
            if (rowl0 + 32 * f < rowsv) {
                const size_t o = ((size_t)b * p.nchunk + chunk) * p.r_pitch + rowg0 + 32 * f;
                p.r0[o] = q0[f];
                if (MODE == SR_NN || MODE == SR_LSE) p.r1[o] = q1[f];
                if (MODE != SR_LSE) p.ri[o] = qi[f];
            }
    }
}

// ------------------------------------------------------------------ launch
#define SR_LDS_BYTES(WN) (SR_NSLOT * SR_SLOT_U4 * 16 + SR_TILE * SR_PLD * 4 + 3 * SR_TILE * 4 + SR_MAXLIST * 2)

// column chunks per row block: enough workgroups for two rounds of the 256 CUs of the target part (a function of the sizes alone, so
// that a workspace carved without a handle and the launch agree)
int simred_chunks(int batch, int M, int N) {
    const int nrb = (M + SR_TILE - 1) / SR_TILE, nct = (N + SR_TILE - 1) / SR_TILE;
    int nchunk = (512 + batch * nrb - 1) / (batch * nrb);
    if (nchunk > nct) nchunk = nct;
    if (nchunk < 1) nchunk = 1;
    while ((nct + nchunk - 1) / nchunk > SR_MAXLIST) ++nchunk;
    return nchunk;
}

template <int KS, int WN, int MODE, bool F32>
static int sr_launch_one(imcui_hip_s* h, const SimRedP& p, hipStream_t stream) {
    static std::atomic<unsigned long long> optin{0};  // > 64 KB of dynamic LDS: once per instantiation AND device (common.h)
    constexpr int lds = SR_LDS_BYTES(WN);
    auto kern = simred_kernel<KS, WN, MODE, F32>;
    if (!imcui_lds_optin(optin, reinterpret_cast<const void*>(kern), lds)) return imcui_set_err(h, IMCUI_ERR_HIP, "simred: cannot reserve %d bytes of LDS", lds);
    const int nrb = (p.M + SR_TILE - 1) / SR_TILE;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.batch * p.nchunk * nrb)), dim3(512), lds, stream, p);
    return IMCUI_OK;
}
template <int MODE, bool F32>
static int sr_launch_k(imcui_hip_s* h, const SimRedP& p, hipStream_t stream) {
    switch (p.K) {
        // (512 threads, one workgroup per CU, for every width: the four-slot ring + the parked tile are 110 KB of LDS; the 256-thread
        // geometry WN = 1 -- two workgroups per CU, register transport -- was the first version for K <= 128 and stays instantiable)
        // (K <= 128: 64-row wave tiles, half the LDS reads per matrix instruction; IMCUI_SR_DBG bit 3 = the 32-row geometry everywhere, A/B)
        case 64: return (p.dbg & 8) ? sr_launch_one<4, 2, MODE, F32>(h, p, stream) : sr_launch_one<4, 4, MODE, F32>(h, p, stream);
        case 128: return (p.dbg & 8) ? sr_launch_one<8, 2, MODE, F32>(h, p, stream) : sr_launch_one<8, 4, MODE, F32>(h, p, stream);
        case 256: return sr_launch_one<16, 2, MODE, F32>(h, p, stream);
        default: return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "simred: K=%d (64, 128 or 256)", p.K);
    }
}
template <bool F32>
static int sr_launch_m(imcui_hip_s* h, const SimRedP& p, hipStream_t stream) {
    switch (p.mode) {
        case SR_NN: return sr_launch_k<SR_NN, F32>(h, p, stream);
        case SR_NN1: return sr_launch_k<SR_NN1, F32>(h, p, stream);
        case SR_LSE: return sr_launch_k<SR_LSE, F32>(h, p, stream);
        case SR_DSBEST: return sr_launch_k<SR_DSBEST, F32>(h, p, stream);
        case SR_LGBEST: return sr_launch_k<SR_LGBEST, F32>(h, p, stream);
        default: return imcui_set_err(h, IMCUI_ERR_ARG, "simred: bad mode %d", p.mode);
    }
}

int simred_launch(imcui_hip_s* h, const SimRedP& p, hipStream_t stream) {
    if (p.batch <= 0 || p.M <= 0 || p.N <= 0) return IMCUI_OK;
    if (!p.Ap || !p.Bp || !p.r0 || !p.c0 || p.nchunk < 1 || p.r_pitch < p.M || p.c_pitch < p.N)
        return imcui_set_err(h, IMCUI_ERR_ARG, "simred: null operand / output or pitch below the matrix size");
    const int nct = (p.N + SR_TILE - 1) / SR_TILE;
    if ((nct + p.nchunk - 1) / p.nchunk > SR_MAXLIST) return imcui_set_err(h, IMCUI_ERR_ARG, "simred: more than %d column tiles per chunk", SR_MAXLIST);
    if ((p.mode == SR_DSBEST || p.mode == SR_LGBEST) && (!p.rmax || !p.rsum || !p.cmax || !p.csum || (p.mode == SR_LGBEST && (!p.l0 || !p.l1))))
        return imcui_set_err(h, IMCUI_ERR_ARG, "simred: pass 2 needs the statistics of pass 1");
    if ((p.mode == SR_NN || p.mode == SR_NN1) && p.alpha != 1.0f) return imcui_set_err(h, IMCUI_ERR_ARG, "simred: the nearest-neighbour modes take alpha = 1");
    if (!(p.alpha > 0.0f)) return imcui_set_err(h, IMCUI_ERR_ARG, "simred: alpha must be positive");
    static const int dbg = getenv("IMCUI_SR_DBG") ? atoi(getenv("IMCUI_SR_DBG")) : 0;  // (lab switch, read once per process)
    SimRedP q = p;
    q.dbg = dbg;
    imcui_prof_begin(h, PROF_GEMM, stream);
    const int rc = (h->precision == 1) ? sr_launch_m<false>(h, q, stream) : sr_launch_m<true>(h, q, stream);
    imcui_prof_end(h, PROF_GEMM, stream);
    if (rc != IMCUI_OK) return rc;
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}

// ------------------------------------------------------------------ dual-softmax coarse matching on top of it
// merge `np` partial (max, sum) pairs per item in ascending partial order: out[b][t] over pm / ps [b][np][n]
__global__ void sr_lse_merge_kernel(const float* __restrict__ pm, const float* __restrict__ ps, int np, int n, float* __restrict__ om, float* __restrict__ os) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float m = -INFINITY;
    for (int q = 0; q < np; ++q) m = fmaxf(m, pm[((size_t)b * np + q) * n + t]);
    float s = 0.0f;
    for (int q = 0; q < np; ++q) {
        const float v = pm[((size_t)b * np + q) * n + t];
        if (v > -INFINITY) s += ps[((size_t)b * np + q) * n + t] * __expf(v - m);
    }
    om[(size_t)b * n + t] = m;
    os[(size_t)b * n + t] = s;
}
// Which tiles can hold a confidence above thr?  conf(i, j) <= min(p_col, p_row); inside tile (rb, ct) every x(i, j) <= cpm[rb][j] (the
// column maximum over the block's rows, from pass 1), so an entry above the threshold needs, for its column j,
//   exp(cpm[rb][j] - cmax[j]) / csum[j] > thr   and   exp(cpm[rb][j] - min_i (rmax[i] + log rsum[i])) > thr    (i over the block's rows).
// Evaluated with a 2 % margin (the approximate exp / reciprocal of pass 2 are ~1e-6 relative).  grid (nct, nrb, B), 128 threads.
__global__ __launch_bounds__(128) void sr_ds_flag_kernel(const float* __restrict__ cpm, const float* __restrict__ rmax, const float* __restrict__ rsum,
                                                         const float* __restrict__ cmax, const float* __restrict__ csum, int L, int S, int nrb, int nct, float thr,
                                                         unsigned char* __restrict__ flags) {
    __shared__ float red[2];
    __shared__ int any[2];
    const int ct = blockIdx.x, rb = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int i = rb * SR_TILE + tid;
    float lse = INFINITY;
    if (i < L) lse = rmax[(size_t)b * L + i] + logf(rsum[(size_t)b * L + i]);
    lse = -wave_max(-lse);
    if ((tid & 63) == 0) red[tid >> 6] = lse;
    __syncthreads();
    const float minr = fminf(red[0], red[1]);
    const int j = ct * SR_TILE + tid;
    bool f = false;
    if (j < S) {
        const float v = cpm[((size_t)b * nrb + rb) * S + j];
        const float t = 0.98f * thr;
        f = t <= 0.0f || (expf(v - cmax[(size_t)b * S + j]) > t * csum[(size_t)b * S + j] && expf(v - minr) > t);
    }
    const unsigned long long bal = __ballot(f);
    if ((tid & 63) == 0) any[tid >> 6] = bal != 0ull;
    __syncthreads();
    if (tid == 0) flags[((size_t)b * nrb + rb) * nct + ct] = (any[0] | any[1]) ? 1 : 0;
}
// row best over the column chunks (ascending: a later chunk wins only with a strictly larger value = first column attaining the maximum)
__global__ void sr_rowbest_merge_kernel(const float* __restrict__ rpv, const int* __restrict__ rpj, int nch, int L, float* __restrict__ best, int* __restrict__ bestj) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    float bv = -1.0f;
    int bj = 0x7fffffff;
    for (int c = 0; c < nch; ++c) {
        const float v = rpv[((size_t)b * nch + c) * L + i];
        if (v > bv) {
            bv = v;
            bj = rpj[((size_t)b * nch + c) * L + i];
        }
    }
    best[(size_t)b * L + i] = bv;
    bestj[(size_t)b * L + i] = bj;
}
// column best over the row blocks whose tile was evaluated
__global__ void sr_colbest_merge_kernel(const float* __restrict__ cpv, const unsigned char* __restrict__ flags, int nrb, int nct, int S, float* __restrict__ cbest) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= S) return;
    const int ct = j / SR_TILE;
    float m = -1.0f;
    for (int q = 0; q < nrb; ++q)
        if (flags[((size_t)b * nrb + q) * nct + ct]) m = fmaxf(m, cpv[((size_t)b * nrb + q) * S + j]);
    cbest[(size_t)b * S + j] = m;
}

void simred_ds_carve(WsAlloc& a, int B, int L, int S, int K, SimDsWs& w) {
    w.nrb = (L + SR_TILE - 1) / SR_TILE;
    w.nct = (S + SR_TILE - 1) / SR_TILE;
    w.nchunk = simred_chunks(B, L, S);
    w.ap = a.get<uint4>((size_t)B * simred_packed_uint4(L, K));
    w.bp = a.get<uint4>((size_t)B * simred_packed_uint4(S, K));
    w.rp0 = a.get<float>((size_t)B * w.nchunk * L);
    w.rp1 = a.get<float>((size_t)B * w.nchunk * L);
    w.rpj = a.get<int>((size_t)B * w.nchunk * L);
    w.cp0 = a.get<float>((size_t)B * w.nrb * S);
    w.cp1 = a.get<float>((size_t)B * w.nrb * S);
    w.flags = a.get<unsigned char>((size_t)B * w.nrb * w.nct);
}

int simred_dual_softmax(imcui_hip_s* h, const SimDsWs& w, const float* fa, long lda, long a_bs, const float* fb, long ldb, long b_bs, int B, int L, int S, int K,
                        float alpha, float thr, float* rmax, float* rsum, float* cmax, float* csum, float* best, int* bestj, float* cbest, hipStream_t stream) {
    if (!simred_ok(K)) return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "dual-softmax: feature width %d (64, 128 or 256)", K);
    if (h->precision == 1 && h->range_flag) {  // opt-in debugging aid: scan the f32 operands that are about to be split (a negative row stride = rows in reverse)
        for (int z = 0; z < B; ++z) {
            imcui_range_check(h, fa + (size_t)z * a_bs + (lda < 0 ? (long)(L - 1) * lda : 0), L, K, lda < 0 ? -lda : lda, nullptr, 0, stream);
            imcui_range_check(h, fb + (size_t)z * b_bs + (ldb < 0 ? (long)(S - 1) * ldb : 0), S, K, ldb < 0 ? -ldb : ldb, nullptr, 0, stream);
        }
    }
    simred_pack(h, fa, lda, 1, a_bs, L, K, B, nullptr, 0, w.ap, stream);
    simred_pack(h, fb, ldb, 1, b_bs, S, K, B, nullptr, 0, w.bp, stream);
    SimRedP p;
    p.mode = SR_LSE;
    p.Ap = w.ap, p.Bp = w.bp;
    p.ap_bs = (long)simred_packed_uint4(L, K), p.bp_bs = (long)simred_packed_uint4(S, K);
    p.M = L, p.N = S, p.K = K, p.batch = B;
    p.alpha = alpha;
    p.nchunk = w.nchunk;
    p.r0 = w.rp0, p.r1 = w.rp1, p.ri = w.rpj, p.r_pitch = L;
    p.c0 = w.cp0, p.c1 = w.cp1, p.c_pitch = S;
    int rc = simred_launch(h, p, stream);
    if (rc != IMCUI_OK) return rc;
    const dim3 blk(256);
    hipLaunchKernelGGL(sr_lse_merge_kernel, dim3((L + 255) / 256, B), blk, 0, stream, w.rp0, w.rp1, w.nchunk, L, rmax, rsum);
    hipLaunchKernelGGL(sr_lse_merge_kernel, dim3((S + 255) / 256, B), blk, 0, stream, w.cp0, w.cp1, w.nrb, S, cmax, csum);
    hipLaunchKernelGGL(sr_ds_flag_kernel, dim3(w.nct, w.nrb, B), dim3(128), 0, stream, w.cp0, rmax, rsum, cmax, csum, L, S, w.nrb, w.nct, thr, w.flags);
    p.mode = SR_DSBEST;
    p.rmax = rmax, p.rsum = rsum, p.cmax = cmax, p.csum = csum;
    p.flags = w.flags;
    rc = simred_launch(h, p, stream);
    if (rc != IMCUI_OK) return rc;
    hipLaunchKernelGGL(sr_rowbest_merge_kernel, dim3((L + 255) / 256, B), blk, 0, stream, w.rp0, w.rpj, w.nchunk, L, best, bestj);
    hipLaunchKernelGGL(sr_colbest_merge_kernel, dim3((S + 255) / 256, B), blk, 0, stream, w.cp0, w.flags, w.nrb, w.nct, S, cbest);
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}

// ------------------------------------------------------------------ test hook (tests/test_gpu_simred.py): ONE launch on raw matrices
// A [batch][M][K], Bm [batch][N][K] f32 (device); mcnt / ncnt [batch] device-side sizes or null; nchunk 0 = simred_chunks().
// Row outputs [batch][nchunk][M], column outputs [batch][ceil(M / 128)][N]; which of them a mode writes: SimRedP (simred.h).
extern "C" int imcui_hip_simred_chunks(int batch, int M, int N) { return simred_chunks(batch, M, N); }
extern "C" size_t imcui_hip_simred_debug_workspace_bytes(int batch, int M, int N, int K) {
    WsAlloc a(nullptr, 0);
    a.get<uint4>((size_t)batch * simred_packed_uint4(M, K));
    a.get<uint4>((size_t)batch * simred_packed_uint4(N, K));
    return a.off;
}
extern "C" int imcui_hip_simred_debug(imcui_hip_s* h, int mode, const float* A, const float* Bm, int batch, int M, int N, int K, const int* mcnt, const int* ncnt,
                                      float alpha, int nchunk, float* r0, float* r1, int* ri, float* c0, float* c1, int* ci, const float* rmax, const float* rsum,
                                      const float* cmax, const float* csum, const float* l0, const float* l1, const unsigned char* flags, void* ws, size_t ws_bytes,
                                      void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!h || !A || !Bm) return IMCUI_ERR_ARG;
    if (!simred_ok(K)) return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "simred: K=%d (64, 128 or 256)", K);
    WsAlloc a(ws, ws_bytes);
    uint4* ap = a.get<uint4>((size_t)batch * simred_packed_uint4(M, K));
    uint4* bp = a.get<uint4>((size_t)batch * simred_packed_uint4(N, K));
    if (!ws || !a.ok) return imcui_set_err(h, IMCUI_ERR_WS, "simred: workspace too small (%zu bytes)", ws_bytes);
    simred_pack(h, A, K, 1, (long)M * K, M, K, batch, mcnt, 1, ap, stream);
    simred_pack(h, Bm, K, 1, (long)N * K, N, K, batch, ncnt, 1, bp, stream);
    SimRedP p;
    p.mode = mode;
    p.Ap = ap, p.Bp = bp;
    p.ap_bs = (long)simred_packed_uint4(M, K), p.bp_bs = (long)simred_packed_uint4(N, K);
    p.M = M, p.N = N, p.K = K, p.batch = batch;
    p.mcnt = mcnt, p.ncnt = ncnt, p.cnt_stride = 1;
    p.alpha = alpha;
    p.nchunk = nchunk > 0 ? nchunk : simred_chunks(batch, M, N);
    p.r0 = r0, p.r1 = r1, p.ri = ri, p.r_pitch = M;
    p.c0 = c0, p.c1 = c1, p.ci = ci, p.c_pitch = N;
    p.rmax = rmax, p.rsum = rsum, p.cmax = cmax, p.csum = csum;
    p.l0 = l0, p.l1 = l1, p.l0_bs = M, p.l1_bs = N;
    p.flags = flags;
    return simred_launch(h, p, stream);
}
