// SuperPoint forward on MI355X: VGG encoder + detector / descriptor heads + NMS + ordered
// key-point selection + descriptor sampling.  Replaces the `self.net(data, self.conf)` call of
// imcui/hloc/extractors/superpoint.py:56-57.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "conv.h"
#include "gemm.h"
#include "imcui_hip.h"

// ------------------------------------------------------------------ packed weight layout
struct SpLayout {
    size_t w[12], b[12];
    size_t wh[12], wl[12], ws[12];  // split planes (f16 hi / lo, as float offsets) + 2^-e scale
    size_t total;
};
static const int SP_COUT[12] = {64, 64, 64, 64, 128, 128, 128, 128, 256, 65, 256, 256};
static const int SP_CIN[12] = {1, 64, 64, 64, 64, 128, 128, 128, 128, 256, 128, 256};
static const int SP_K[12] = {3, 3, 3, 3, 3, 3, 3, 3, 3, 1, 3, 1};
enum { L1A, L1B, L2A, L2B, L3A, L3B, L4A, L4B, LPA, LPB, LDA, LDB };

static SpLayout sp_layout() {
    SpLayout l;
    size_t off = 0;
    for (int i = 0; i < 12; ++i) {
        l.w[i] = off;
        off += align_up((size_t)SP_COUT[i] * SP_CIN[i] * SP_K[i] * SP_K[i], 64);
        l.b[i] = off;
        off += align_up((size_t)SP_COUT[i], 64);
    }
    for (int i = 1; i < 12; ++i) {  // conv1a (Cin = 1) stays on the VALU
        // 1x1 layers are stored fragment-major with rows padded to a multiple of 32 (convPb: 65 -> 96)
        const size_t n = (size_t)(SP_K[i] == 1 ? align_up(SP_COUT[i], 32) : SP_COUT[i]) * SP_CIN[i] * SP_K[i] * SP_K[i];
        l.wh[i] = off;
        off += align_up(n / 2 + 1, 64);
        l.wl[i] = off;
        off += align_up(n / 2 + 1, 64);
        l.ws[i] = off;
        off += 64;
    }
    l.total = off;
    return l;
}

extern "C" size_t imcui_hip_superpoint_packed_floats(void) { return sp_layout().total; }

extern "C" int imcui_hip_superpoint_pack_weights(const float* const* w, const float* const* b, float* packed) {
    if (!w || !b || !packed) return IMCUI_ERR_ARG;
    const SpLayout l = sp_layout();
    memset(packed, 0, l.total * sizeof(float));
    for (int i = 0; i < 12; ++i) {
        if (!w[i] || !b[i]) return IMCUI_ERR_ARG;
        if (i == L1A)
            pack_conv1a(w[i], packed + l.w[i]);
        else if (SP_K[i] == 3)
            pack_conv3x3(w[i], SP_COUT[i], SP_CIN[i], packed + l.w[i]);
        else  // 1x1: [Cout][Cin] is already the K-contiguous GEMM layout
            memcpy(packed + l.w[i], w[i], (size_t)SP_COUT[i] * SP_CIN[i] * sizeof(float));
        memcpy(packed + l.b[i], b[i], (size_t)SP_COUT[i] * sizeof(float));
        if (i == L1A) continue;
        unsigned short* hp = reinterpret_cast<unsigned short*>(packed + l.wh[i]);
        unsigned short* lp = reinterpret_cast<unsigned short*>(packed + l.wl[i]);
        if (SP_K[i] == 3)
            packed[l.ws[i]] = pack_conv3x3_split(w[i], SP_COUT[i], SP_CIN[i], hp, lp);
        else
            packed[l.ws[i]] = split_weights_frag_host(w[i], SP_COUT[i], SP_CIN[i], hp, lp);
    }
    return IMCUI_OK;
}

// ------------------------------------------------------------------ detector soft-max + depth-to-space
// One wave per coarse cell: lane l holds logit l, the dustbin (65th) logit is broadcast.
__global__ __launch_bounds__(256) void sp_softmax_kernel(const float* __restrict__ logits, int ldl,
                                                         float* __restrict__ scores, int Hc, int Wc, long ncell) {
    const int lane = threadIdx.x & 63;
    const long cell = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= ncell) return;
    const float* row = logits + cell * ldl;
    const float x = row[lane];
    const float dust = row[64];
    const float m = fmaxf(wave_max(x), dust);
    const float e = expf(x - m);
    const float sum = wave_sum(e) + expf(dust - m);
    const float pr = e / sum;
    const int cx = (int)(cell % Wc);
    const long t = cell / Wc;
    const int cy = (int)(t % Hc);
    const long b = t / Hc;
    const int W = Wc * 8, H = Hc * 8;
    scores[(b * H + cy * 8 + (lane >> 3)) * W + cx * 8 + (lane & 7)] = pr;
}

// ------------------------------------------------------------------ fused simple_nms
// All five (2r+1)^2 max-pools of upstream `simple_nms` in one pass over an LDS tile with a
// 5r halo; each pool is separable (row max then column max), and max is exact in any order, so
// the result is bitwise the reference's.  Out-of-image = -inf (scores) / 0 (masks), i.e.
// max_pool2d's implicit padding.  Every thread produces runs of 8 outputs from a register window
// (8 + 2r LDS reads instead of 8 * (2r+1)); the LDS row stride is odd so both the row pass
// (lanes along rows) and the column pass (lanes along columns) are bank-conflict free.
// LDS per workgroup (round 4): the scores and ONE float temporary (the row-pass result) + the two masks as bytes; the suppressed scores
// `where(supp_mask, 0, scores)` are formed while the row pass reads them instead of being stored.  53 KB at r = 4 (it was 105 KB with
// five float planes), i.e. three workgroups per CU instead of one: the kernel is a chain of ten barrier-separated passes of dependent
// LDS reads, and with one wave per SIMD nothing covered their latency.
#define NMS_T 32
template <int R>
__global__ __launch_bounds__(256) void sp_nms_kernel(const float* __restrict__ in, float* __restrict__ out, int H,
                                                     int W) {
    extern __shared__ __attribute__((aligned(16))) char nms_smem[];
    constexpr int halo = 5 * R;
    constexpr int Rg = NMS_T + 2 * halo;
    constexpr int RS = Rg | 1;               // odd row stride (floats)
    constexpr int RB = ((Rg + 3) / 4) | 1;   // row stride of the byte planes in 4-byte words, odd: lanes along rows hit distinct banks
    constexpr int RSB = RB * 4;
    constexpr int n = Rg * RS;
    float* S = reinterpret_cast<float*>(nms_smem);  // scores (-inf outside the image)
    float* T = S + n;                               // row-pass temporary
    unsigned char* M = reinterpret_cast<unsigned char*>(T + n);  // max_mask (0/1)
    unsigned char* U = M + Rg * RSB;                             // supp_mask (0/1; 0 outside the image)
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int y0 = blockIdx.y * NMS_T - halo, x0 = blockIdx.x * NMS_T - halo;
    const float* img = in + (size_t)b * H * W;

    for (int i = tid; i < Rg * Rg; i += 256) {
        const int u = i / Rg, v = i - u * Rg;
        const int y = y0 + u, x = x0 + v;
        S[u * RS + v] = (y >= 0 && y < H && x >= 0 && x < W) ? img[(size_t)y * W + x] : -INFINITY;
    }
    __syncthreads();

    // Each pool only has to be right where a later stage still reads it: pool p (1..5) is evaluated on the
    // tile plus a halo of (5 - p) R, i.e. with a margin of m = p R from the staged region (the last pool
    // exactly on the 32 x 32 core).  That halves the work of evaluating all five on the full 5R-halo region.
    // T = max over [v-R, v+R] of src(u, v) (row pass) for rows [m-R, Rg-m+R) x columns [m, Rg-m).
    // work item = (row u, chunk of 8 columns); lanes run along u.
    auto rowpass = [&](auto&& src, auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        constexpr int nu = Rg - 2 * (m - R), nv = Rg - 2 * m, nch = (nv + 7) / 8;
        for (int it = tid; it < nu * nch; it += 256) {
            const int u = (m - R) + it % nu, v0 = m + (it / nu) * 8;
            float w[8 + 2 * R];
#pragma unroll
            for (int j = 0; j < 8 + 2 * R; ++j) {
                const int v = v0 - R + j;
                w[j] = (v < Rg) ? src(u, v) : -INFINITY;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float mx = w[j];
#pragma unroll
                for (int k = 1; k <= 2 * R; ++k) mx = fmaxf(mx, w[j + k]);
                if (v0 + j < Rg - m) T[u * RS + v0 + j] = mx;
            }
        }
    };
    // column pass over T on [m, Rg-m)^2; calls f(u, v, pooled value).  work item = (chunk of 8 rows, column v);
    // lanes along v.
    auto colpass = [&](auto mc, auto&& f) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        constexpr int nv = Rg - 2 * m, nch = (nv + 7) / 8;
        for (int it = tid; it < nv * nch; it += 256) {
            const int v = m + it % nv, u0 = m + (it / nv) * 8;
            float w[8 + 2 * R];
#pragma unroll
            for (int j = 0; j < 8 + 2 * R; ++j) {
                const int u = u0 - R + j;
                w[j] = (u < Rg) ? T[u * RS + v] : -INFINITY;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float mx = w[j];
#pragma unroll
                for (int k = 1; k <= 2 * R; ++k) mx = fmaxf(mx, w[j + k]);
                if (u0 + j < Rg - m) f(u0 + j, v, mx);
            }
        }
    };
    auto inimg = [&](int u, int v) __attribute__((always_inline)) {
        const int y = y0 + u, x = x0 + v;
        return y >= 0 && y < H && x >= 0 && x < W;
    };
    auto score = [&](int u, int v) __attribute__((always_inline)) { return S[u * RS + v]; };
    auto maxmask = [&](int u, int v) __attribute__((always_inline)) { return (float)M[u * RSB + v]; };
    // supp_scores = where(supp_mask, 0, scores), -inf outside the image (S is -inf and U is 0 there)
    auto suppscore = [&](int u, int v) __attribute__((always_inline)) { return U[u * RSB + v] ? 0.0f : S[u * RS + v]; };
    using M1 = std::integral_constant<int, 1 * R>;
    using M2 = std::integral_constant<int, 2 * R>;
    using M3 = std::integral_constant<int, 3 * R>;
    using M4 = std::integral_constant<int, 4 * R>;
    using M5 = std::integral_constant<int, 5 * R>;

    // max_mask = scores == max_pool(scores)
    rowpass(score, M1{});
    __syncthreads();
    colpass(M1{}, [&](int u, int v, float pm) { M[u * RSB + v] = (inimg(u, v) && S[u * RS + v] == pm) ? 1 : 0; });
    __syncthreads();
    auto iteration = [&](auto ma, auto mb) __attribute__((always_inline)) {
        // supp_mask = max_pool(max_mask) > 0
        rowpass(maxmask, ma);
        __syncthreads();
        colpass(ma, [&](int u, int v, float pm) { U[u * RSB + v] = (pm > 0.0f && inimg(u, v)) ? 1 : 0; });
        __syncthreads();
        // new_max_mask = supp_scores == max_pool(supp_scores) ; max_mask |= new_max_mask & ~supp_mask
        rowpass(suppscore, mb);
        __syncthreads();
        colpass(mb, [&](int u, int v, float pm) {
            // (not suppressed => supp_scores = scores at this pixel)
            if (inimg(u, v) && U[u * RSB + v] == 0 && S[u * RS + v] == pm) M[u * RSB + v] = 1;
        });
        __syncthreads();
    };
    iteration(M2{}, M3{});
    iteration(M4{}, M5{});
    float* dst = out + (size_t)b * H * W;
    for (int i = tid; i < NMS_T * NMS_T; i += 256) {
        const int ty = i / NMS_T, tx = i - ty * NMS_T;
        const int y = blockIdx.y * NMS_T + ty, x = blockIdx.x * NMS_T + tx;
        if (y < H && x < W) dst[(size_t)y * W + x] = M[(ty + halo) * RSB + tx + halo] ? S[(ty + halo) * RS + tx + halo] : 0.0f;
    }
}

template <int R>
static void nms_launch_r(const float* in, float* out, int B, int H, int W, hipStream_t stream) {
    constexpr int Rg = NMS_T + 10 * R;
    constexpr size_t smem = (size_t)Rg * (Rg | 1) * 2 * sizeof(float) + (size_t)2 * Rg * 4 * (((Rg + 3) / 4) | 1);  // two float planes + two byte planes
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sp_nms_kernel<R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    dim3 grid(cdiv(W, NMS_T), cdiv(H, NMS_T), B);
    hipLaunchKernelGGL(sp_nms_kernel<R>, grid, dim3(256), smem, stream, in, out, H, W);
}

static int nms_launch(imcui_hip_s* h, const float* in, float* out, int B, int H, int W, int r, hipStream_t stream) {
    switch (r) {
        case 0: nms_launch_r<0>(in, out, B, H, W, stream); break;
        case 1: nms_launch_r<1>(in, out, B, H, W, stream); break;
        case 2: nms_launch_r<2>(in, out, B, H, W, stream); break;
        case 3: nms_launch_r<3>(in, out, B, H, W, stream); break;
        case 4: nms_launch_r<4>(in, out, B, H, W, stream); break;
        default: return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "nms_radius=%d not supported (0..4)", r);
    }
    IMCUI_CHECK_LAUNCH(h);
    return IMCUI_OK;
}

// ------------------------------------------------------------------ ordered candidate compaction
// candidate: nms score > thr and inside the border band.  Row-major order is preserved (the
// reference returns `nonzero` order when no top-k is needed).
#define SEL_CHUNK 4096  // pixels per block (16 consecutive per thread)

__device__ __forceinline__ bool sp_is_cand(float s, int idx, int H, int W, float thr, int border) {
    const int y = idx / W, x = idx - y * W;
    return s > thr && y >= border && y < H - border && x >= border && x < W - border;
}

__global__ __launch_bounds__(256) void sp_count_kernel(const float* __restrict__ nms, int H, int W, float thr,
                                                       int border, int* __restrict__ blkcnt, int nchunk) {
    __shared__ int wsum[4];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int npix = H * W;
    const float* img = nms + (size_t)b * npix;
    const int base = chunk * SEL_CHUNK + threadIdx.x * 16;
    int c = 0;
    for (int j = 0; j < 16; ++j) {
        const int idx = base + j;
        if (idx < npix && sp_is_cand(img[idx], idx, H, W, thr, border)) ++c;
    }
    c = wave_sum_i(c);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blkcnt[b * nchunk + chunk] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the per-chunk counts (one block per image, serial over <= a few hundred chunks)
__global__ void sp_scan_kernel(const int* __restrict__ blkcnt, int* __restrict__ blkoff, int* __restrict__ ncand,
                               int nchunk) {
    if (threadIdx.x != 0) return;
    const int b = blockIdx.x;
    int run = 0;
    for (int i = 0; i < nchunk; ++i) {
        blkoff[b * nchunk + i] = run;
        run += blkcnt[b * nchunk + i];
    }
    ncand[b] = run;
}

__global__ __launch_bounds__(256) void sp_compact_kernel(const float* __restrict__ nms, int H, int W, float thr,
                                                         int border, const int* __restrict__ blkoff, int nchunk,
                                                         unsigned long long* __restrict__ cand, int cand_cap) {
    __shared__ int tcnt[256];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int npix = H * W;
    const float* img = nms + (size_t)b * npix;
    const int base = chunk * SEL_CHUNK + threadIdx.x * 16;
    float v[16];
    unsigned flags = 0;
    int c = 0;
    for (int j = 0; j < 16; ++j) {
        const int idx = base + j;
        v[j] = (idx < npix) ? img[idx] : 0.0f;
        if (idx < npix && sp_is_cand(v[j], idx, H, W, thr, border)) {
            flags |= 1u << j;
            ++c;
        }
    }
    tcnt[threadIdx.x] = c;
    __syncthreads();
    // exclusive prefix over the 256 per-thread counts (Hillis-Steele)
    for (int o = 1; o < 256; o <<= 1) {
        int add = (threadIdx.x >= o) ? tcnt[threadIdx.x - o] : 0;
        __syncthreads();
        tcnt[threadIdx.x] += add;
        __syncthreads();
    }
    int pos = blkoff[b * nchunk + chunk] + tcnt[threadIdx.x] - c;
    unsigned long long* dst = cand + (size_t)b * cand_cap;
    for (int j = 0; j < 16; ++j)
        if (flags & (1u << j)) {
            const unsigned idx = (unsigned)(base + j);
            if (pos < cand_cap)
                dst[pos] = ((unsigned long long)__float_as_uint(v[j]) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
            ++pos;
        }
}

// ------------------------------------------------------------------ top-k (score desc, index asc)
// One block per image.  64-bit keys (score bits << 32 | ~index) are unique, so "the k largest
// keys" is a well defined set: 8 rounds of 8-bit radix select find the k-th key, the winners are
// gathered into LDS and bitonic-sorted descending.
// TOPK_MAX keys of 8 bytes = 128 KiB of the CU's 160 KiB LDS (dynamic allocation; the UI's max_keypoints slider
// ends at 10000, imcui/ui/app_class.py).  Larger k is rejected on the host before the launch.
#define TOPK_MAX 16384
__global__ __launch_bounds__(1024) void sp_topk_kernel(const unsigned long long* __restrict__ cand, int cand_cap,
                                                       const int* __restrict__ ncand, int max_kpts, int kcap, int W,
                                                       float* __restrict__ kpts, float* __restrict__ scores,
                                                       int* __restrict__ nkpts, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sel[];  // [roundup_pow2(min(max_kpts, TOPK_MAX))]
    __shared__ int hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_k, s_nsel;
    const int b = blockIdx.x, tid = threadIdx.x;
    const unsigned long long* c = cand + (size_t)b * cand_cap;
    int n = ncand[b];
    if (n > cand_cap) {
        if (tid == 0) atomicOr(status, 1);  // candidate buffer overflow
        n = cand_cap;
    }
    int k = (max_kpts < 0) ? n : min(n, max_kpts);
    float* kp = kpts + (size_t)b * kcap * 2;
    float* sc = scores + (size_t)b * kcap;
    if (k > kcap) {
        if (tid == 0) atomicOr(status, 2);  // output capacity too small
        k = kcap;
    }
    if (tid == 0) nkpts[b] = k;
    if (n <= k || max_kpts < 0) {
        // no top-k: keep `nonzero` (row-major) order
        for (int i = tid; i < k; i += 1024) {
            const unsigned long long key = c[i];
            const unsigned idx = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
            kp[2 * i + 0] = (float)(idx % W);
            kp[2 * i + 1] = (float)(idx / W);
            sc[i] = __uint_as_float((unsigned)(key >> 32));
        }
        return;
    }
    if (k > TOPK_MAX) {
        if (tid == 0) {
            atomicOr(status, 4);  // top-k larger than the LDS sorter supports
            nkpts[b] = 0;
        }
        return;
    }
    // ---- radix select: find the k-th largest key
    if (tid == 0) {
        s_prefix = 0;
        s_k = k;
    }
    __syncthreads();
    for (int byte = 7; byte >= 0; --byte) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        const unsigned long long himask = (byte == 7) ? 0ull : (~0ull << (8 * (byte + 1)));
        for (int i = tid; i < n; i += 1024) {
            const unsigned long long key = c[i];
            if ((key & himask) == prefix) atomicAdd(&hist[(int)((key >> (8 * byte)) & 0xFF)], 1);
        }
        __syncthreads();
        // the digit of the k-th key: walk the bins downwards until the running count reaches k.  One wave does it in parallel (lane l owns
        // bins 255 - 4 l .. 252 - 4 l, inclusive scan over the lanes, the first lane that reaches k finishes inside its four bins); one
        // thread walking 255 dependent LDS reads, eight times, was most of this kernel's 84 us
        if (tid < 64) {
            const int d0 = 255 - 4 * tid;
            const int h0 = hist[d0], h1 = hist[d0 - 1], h2 = hist[d0 - 2], h3 = hist[d0 - 3];
            const int mine = h0 + h1 + h2 + h3;
            int inc = mine;
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(inc, o, 64);
                if (tid >= o) inc += up;
            }
            const int kk = s_k;
            const unsigned long long reach = __ballot(inc >= kk);
            const int L = reach ? __ffsll((long long)reach) - 1 : 63;  // (never empty: at least k keys carry the prefix; 63 = bin 0 all the same)
            if (tid == L) {
                int k2 = kk - (inc - mine), d;
                if (h0 >= k2)
                    d = d0;
                else if (h0 + h1 >= k2)
                    d = d0 - 1, k2 -= h0;
                else if (h0 + h1 + h2 >= k2)
                    d = d0 - 2, k2 -= h0 + h1;
                else
                    d = d0 - 3, k2 -= h0 + h1 + h2;  // (lane 63: bin 0 takes what is left, as the sequential walk did)
                s_prefix = prefix | ((unsigned long long)d << (8 * byte));
                s_k = k2;
            }
        }
        __syncthreads();
    }
    const unsigned long long kth = s_prefix;  // exactly k keys are >= kth
    if (tid == 0) s_nsel = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const unsigned long long key = c[i];
        if (key >= kth) {
            const int pos = atomicAdd(&s_nsel, 1);
            if (pos < TOPK_MAX) sel[pos] = key;
        }
    }
    __syncthreads();
    int np2 = 1;
    while (np2 < k) np2 <<= 1;
    for (int i = k + tid; i < np2; i += 1024) sel[i] = 0ull;
    __syncthreads();
    // ---- bitonic sort, descending
    for (int size = 2; size <= np2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < np2 / 2; i += 1024) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi2 = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = sel[lo], bb = sel[hi2];
                if ((a < bb) == desc) {
                    sel[lo] = bb;
                    sel[hi2] = a;
                }
            }
            // pair i of a stage with stride <= 64 lies inside elements [128 w, 128 w + 127] of the wave w that handles it: such stages only
            // need the wave's own LDS traffic to have landed.  A workgroup barrier is needed around the stages that cross waves
            // (stride > 64): 10 of the 66 stages of a 2048-key sort
            const int next = (stride > 1) ? (stride >> 1) : size;  // stride of the following stage
            if (stride > 64 || next > 64) {
                __syncthreads();
            } else {
                __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < k; i += 1024) {
        const unsigned long long key = sel[i];
        const unsigned idx = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
        kp[2 * i + 0] = (float)(idx % W);
        kp[2 * i + 1] = (float)(idx / W);
        sc[i] = __uint_as_float((unsigned)(key >> 32));
    }
}

// ------------------------------------------------------------------ descriptor sampling
// One wave per key-point: bilinear interpolation (grid_sample, align_corners=True, or the
// reference's "fix_sampling" variant) of the L2-normalised dense descriptors, then L2 norm.
// Lane l handles channels 4l..4l+3.  The dense normalisation is applied on the fly to the four
// neighbour rows.
__device__ __forceinline__ float4 sp_norm_row(const float* base, long row, int lane, bool valid) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) v = *reinterpret_cast<const float4*>(base + row * 256 + lane * 4);
    const float nrm = sqrtf(wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w));
    const float d = fmaxf(nrm, 1e-12f);
    v.x /= d;
    v.y /= d;
    v.z /= d;
    v.w /= d;
    return v;
}

__global__ __launch_bounds__(256) void sp_sample_kernel(const float* __restrict__ ddesc, const float* __restrict__ kpts,
                                                        const int* __restrict__ nkpts, int kcap, int Hc, int Wc,
                                                        int fix_sampling, float* __restrict__ desc) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nkpts[b]) return;
    const float kx = kpts[((size_t)b * kcap + i) * 2 + 0];
    const float ky = kpts[((size_t)b * kcap + i) * 2 + 1];
    float ix, iy;
    const float s = 8.0f;
    if (!fix_sampling) {
        // upstream: k = k - s/2 + 0.5 ; k /= [w*s - s/2 - 0.5, h*s - s/2 - 0.5] ; k = k*2 - 1
        float gx = ((kx - s / 2) + 0.5f) / ((float)Wc * s - s / 2 - 0.5f);
        float gy = ((ky - s / 2) + 0.5f) / ((float)Hc * s - s / 2 - 0.5f);
        gx = gx * 2.0f - 1.0f;
        gy = gy * 2.0f - 1.0f;
        // grid_sample unnormalise, align_corners=True: (g + 1) * (size - 1) / 2
        ix = (gx + 1.0f) * ((float)(Wc - 1) * 0.5f);
        iy = (gy + 1.0f) * ((float)(Hc - 1) * 0.5f);
    } else {
        // imcui/hloc/extractors/superpoint.py:16-30: (k + 0.5) / (w*s), align_corners=False
        float gx = (kx + 0.5f) / ((float)Wc * s);
        float gy = (ky + 0.5f) / ((float)Hc * s);
        gx = gx * 2.0f - 1.0f;
        gy = gy * 2.0f - 1.0f;
        ix = ((gx + 1.0f) * (float)Wc - 1.0f) * 0.5f;
        iy = ((gy + 1.0f) * (float)Hc - 1.0f) * 0.5f;
    }
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wnw = ((float)x1 - ix) * ((float)y1 - iy);
    const float wne = (ix - (float)x0) * ((float)y1 - iy);
    const float wsw = ((float)x1 - ix) * (iy - (float)y0);
    const float wse = (ix - (float)x0) * (iy - (float)y0);
    const float* base = ddesc + (size_t)b * Hc * Wc * 256;
    const bool vx0 = x0 >= 0 && x0 < Wc, vx1 = x1 >= 0 && x1 < Wc;
    const bool vy0 = y0 >= 0 && y0 < Hc, vy1 = y1 >= 0 && y1 < Hc;
    const float4 nw = sp_norm_row(base, (long)y0 * Wc + x0, lane, vx0 && vy0);
    const float4 ne = sp_norm_row(base, (long)y0 * Wc + x1, lane, vx1 && vy0);
    const float4 sw = sp_norm_row(base, (long)y1 * Wc + x0, lane, vx0 && vy1);
    const float4 se = sp_norm_row(base, (long)y1 * Wc + x1, lane, vx1 && vy1);
    float4 o;
    o.x = nw.x * wnw + ne.x * wne + sw.x * wsw + se.x * wse;
    o.y = nw.y * wnw + ne.y * wne + sw.y * wsw + se.y * wse;
    o.z = nw.z * wnw + ne.z * wne + sw.z * wsw + se.z * wse;
    o.w = nw.w * wnw + ne.w * wne + sw.w * wsw + se.w * wse;
    const float nrm = sqrtf(wave_sum(o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w));
    const float d = fmaxf(nrm, 1e-12f);
    o.x /= d;
    o.y /= d;
    o.z /= d;
    o.w /= d;
    *reinterpret_cast<float4*>(desc + ((size_t)b * kcap + i) * 256 + lane * 4) = o;
}

// ------------------------------------------------------------------ forward
struct SpWs {
    float *a1a, *p1, *a2a, *p2, *a3a, *p3, *a4a, *feat, *head, *logits, *dense, *nms, *ddesc;
    int *blkcnt, *blkoff, *ncand, *status;
    unsigned long long* cand;
    size_t total;
    bool ok;
};

static SpWs sp_carve(void* ws, size_t ws_bytes, int B, int H, int W, int cand_cap) {
    WsAlloc a(ws, ws_bytes);
    SpWs s;
    const size_t hw = (size_t)B * H * W;
    const size_t c = hw / 64;  // coarse cells
    s.a1a = a.get<float>(hw * 64);
    s.p1 = a.get<float>(hw / 4 * 64);
    s.a2a = a.get<float>(hw / 4 * 64);
    s.p2 = a.get<float>(hw / 16 * 64);
    s.a3a = a.get<float>(hw / 16 * 128);
    s.p3 = a.get<float>(c * 128);
    s.a4a = a.get<float>(c * 128);
    s.feat = a.get<float>(c * 128);
    s.head = a.get<float>(c * 256);
    s.logits = a.get<float>(c * 65 + 64);
    s.dense = a.get<float>(hw);
    s.nms = a.get<float>(hw);
    s.ddesc = a.get<float>(c * 256);
    const int nchunk = cdiv(H * W, SEL_CHUNK);
    s.blkcnt = a.get<int>((size_t)B * nchunk);
    s.blkoff = a.get<int>((size_t)B * nchunk);
    s.ncand = a.get<int>(B);
    s.status = a.get<int>(1);
    s.cand = a.get<unsigned long long>((size_t)B * cand_cap);
    s.total = a.off;
    s.ok = a.ok;
    return s;
}

// Every pixel can be a candidate when scores tie exactly (flat images), so the candidate list
// is sized for the whole image (8 B per pixel).
static int sp_cand_cap(int H, int W, int r) {
    (void)r;
    return H * W;
}

extern "C" size_t imcui_hip_superpoint_workspace_bytes(int B, int H, int W, int nms_radius) {
    return sp_carve(nullptr, 0, B, H, W, sp_cand_cap(H, W, nms_radius)).total;
}

// Number of key-points a non-degenerate image can yield: NMS leaves at most one maximum per
// (r+1)x(r+1) cell unless scores tie exactly.  Callers size `kcap` with it and retry with H*W
// when imcui_hip_superpoint_status reports bit 1.
extern "C" int imcui_hip_superpoint_max_keypoints_bound(int H, int W, int nms_radius) {
    const int r = nms_radius < 0 ? 0 : nms_radius;
    return cdiv(H, r + 1) * cdiv(W, r + 1);
}

extern "C" int imcui_hip_superpoint_forward(imcui_hip_t* h, const float* packed, const float* image, int B, int H, int W,
                                            int nms_radius, float keypoint_threshold, int remove_borders,
                                            int max_keypoints, int fix_sampling, int kcap, float* keypoints,
                                            float* scores, float* descriptors, int* num_keypoints, int* status,
                                            float* score_map, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!h) return IMCUI_ERR_ARG;
    if (B <= 0) return IMCUI_OK;
    if (H % 8 || W % 8 || H < 8 || W < 8)
        return imcui_set_err(h, IMCUI_ERR_ARG, "superpoint: H=%d W=%d must be positive multiples of 8", H, W);
    if (kcap <= 0 || !packed || !image || !keypoints || !scores || !descriptors || !num_keypoints)
        return imcui_set_err(h, IMCUI_ERR_ARG, "superpoint: null argument or kcap<=0");
    if (max_keypoints > TOPK_MAX && (long)H * W > TOPK_MAX)
        return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "superpoint: max_keypoints=%d exceeds the on-chip top-k sorter (%d); use -1 for all key-points",
                             max_keypoints, TOPK_MAX);
    const int cand_cap = sp_cand_cap(H, W, nms_radius);
    SpWs s = sp_carve(ws, ws_bytes, B, H, W, cand_cap);
    if (!ws || !s.ok) return imcui_set_err(h, IMCUI_ERR_WS, "superpoint: workspace too small (%zu < %zu)", ws_bytes, s.total);
    const SpLayout l = sp_layout();
    const float* P = packed;
    int rc;
#define SPRUN(x)              \
    do {                      \
        rc = (x);             \
        if (rc != IMCUI_OK) return rc; \
    } while (0)
    const int Hc = H / 8, Wc = W / 8;
    const bool split = h->precision == 1;
    auto conv = [&](int L, const float* src, float* dst, int hh, int ww, int pool) -> int {
        if (split)
            return conv3x3_split_launch(h, src, reinterpret_cast<const unsigned short*>(P + l.wh[L]),
                                        reinterpret_cast<const unsigned short*>(P + l.wl[L]), P + l.ws[L], P + l.b[L], dst, B,
                                        hh, ww, SP_CIN[L], SP_COUT[L], 1, pool, stream);
        return conv3x3_launch(h, src, P + l.w[L], P + l.b[L], dst, B, hh, ww, SP_CIN[L], SP_COUT[L], 1, pool, stream);
    };
    auto lin = [&](int L, const float* src, float* dst, int ldc) -> int {
        GemmP g;
        g.epi = EPI_BIAS;
        g.A = src;
        g.lda = SP_CIN[L];
        g.W = P + l.w[L];
        g.ldw = SP_CIN[L];
        if (split) {
            g.Wh = reinterpret_cast<const unsigned short*>(P + l.wh[L]);
            g.Wl = reinterpret_cast<const unsigned short*>(P + l.wl[L]);
            g.wscale = P + l.ws[L];
        }
        g.bias = P + l.b[L];
        g.C = dst;
        g.ldc = ldc;
        g.M = B * Hc * Wc;
        g.N = SP_COUT[L];
        g.K = SP_CIN[L];
        return gemm_launch(h, g, stream);
    };
    // a2: encoder
    if (split) {
        // conv1a is evaluated inside conv1b's patch staging: its 64-channel full-resolution output
        // (79 MB / image) never reaches HBM
        SPRUN(conv1ab_fused_split_launch(h, image, P + l.w[L1A], P + l.b[L1A],
                                         reinterpret_cast<const unsigned short*>(P + l.wh[L1B]),
                                         reinterpret_cast<const unsigned short*>(P + l.wl[L1B]), P + l.ws[L1B], P + l.b[L1B],
                                         s.p1, B, H, W, 1, stream));
    } else {
        SPRUN(conv1a_launch(h, image, P + l.w[L1A], P + l.b[L1A], s.a1a, B, H, W, stream));
        SPRUN(conv(L1B, s.a1a, s.p1, H, W, 1));
    }
    SPRUN(conv(L2A, s.p1, s.a2a, H / 2, W / 2, 0));
    SPRUN(conv(L2B, s.a2a, s.p2, H / 2, W / 2, 1));
    SPRUN(conv(L3A, s.p2, s.a3a, H / 4, W / 4, 0));
    SPRUN(conv(L3B, s.a3a, s.p3, H / 4, W / 4, 1));
    SPRUN(conv(L4A, s.p3, s.a4a, Hc, Wc, 0));
    SPRUN(conv(L4B, s.a4a, s.feat, Hc, Wc, 0));
    // a3: detector head -> dense score map
    SPRUN(conv(LPA, s.feat, s.head, Hc, Wc, 0));
    const long ncell = (long)B * Hc * Wc;
    SPRUN(lin(LPB, s.head, s.logits, 65));
    float* dense = score_map ? score_map : s.dense;
    hipLaunchKernelGGL(sp_softmax_kernel, dim3((unsigned)((ncell + 3) / 4)), dim3(256), 0, stream, s.logits, 65, dense, Hc,
                       Wc, ncell);
    IMCUI_CHECK_LAUNCH(h);
    // a4: NMS ; a5: select
    SPRUN(nms_launch(h, dense, s.nms, B, H, W, nms_radius, stream));
    const int nchunk = cdiv(H * W, SEL_CHUNK);
    int* st = status ? status : s.status;
    hipMemsetAsync(st, 0, sizeof(int), stream);
    hipLaunchKernelGGL(sp_count_kernel, dim3(nchunk, B), dim3(256), 0, stream, s.nms, H, W, keypoint_threshold,
                       remove_borders, s.blkcnt, nchunk);
    hipLaunchKernelGGL(sp_scan_kernel, dim3(B), dim3(64), 0, stream, s.blkcnt, s.blkoff, s.ncand, nchunk);
    hipLaunchKernelGGL(sp_compact_kernel, dim3(nchunk, B), dim3(256), 0, stream, s.nms, H, W, keypoint_threshold,
                       remove_borders, s.blkoff, nchunk, s.cand, cand_cap);
    {
        int np2 = 1;
        const int kmax = max_keypoints < 0 ? 1 : (max_keypoints < TOPK_MAX ? max_keypoints : TOPK_MAX);
        while (np2 < kmax) np2 <<= 1;
        const size_t smem = (size_t)np2 * sizeof(unsigned long long);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sp_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(TOPK_MAX * sizeof(unsigned long long)));
        hipLaunchKernelGGL(sp_topk_kernel, dim3(B), dim3(1024), smem, stream, s.cand, cand_cap, s.ncand, max_keypoints, kcap, W,
                           keypoints, scores, num_keypoints, st);
    }
    IMCUI_CHECK_LAUNCH(h);
    // a6: descriptor head + sampling
    SPRUN(conv(LDA, s.feat, s.head, Hc, Wc, 0));
    SPRUN(lin(LDB, s.head, s.ddesc, 256));
    hipLaunchKernelGGL(sp_sample_kernel, dim3(cdiv(kcap, 4), B), dim3(256), 0, stream, s.ddesc, keypoints, num_keypoints,
                       kcap, Hc, Wc, fix_sampling, descriptors);
    IMCUI_CHECK_LAUNCH(h);
#undef SPRUN
    return IMCUI_OK;
}

// status word of the last forward on this workspace (0 = fine; bit0 candidate overflow,
// bit1 output capacity too small, bit2 top-k larger than the sorter supports). Synchronises.
extern "C" int imcui_hip_superpoint_status(imcui_hip_t* h, int B, int H, int W, int nms_radius, void* ws, size_t ws_bytes,
                                           void* stream_) {
    SpWs s = sp_carve(ws, ws_bytes, B, H, W, sp_cand_cap(H, W, nms_radius));
    if (!ws || !s.ok) return imcui_set_err(h, IMCUI_ERR_WS, "superpoint: workspace too small");
    int st = 0;
    if (hipMemcpyAsync(&st, s.status, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream_) != hipSuccess)
        return imcui_set_err(h, IMCUI_ERR_HIP, "status copy failed");
    hipStreamSynchronize((hipStream_t)stream_);
    if (st) return imcui_set_err(h, IMCUI_ERR_UNSUPPORTED, "superpoint: selection status=%d (1 cand overflow, 2 kcap too small, 4 topk>%d)", st, TOPK_MAX);
    return IMCUI_OK;
}

// stand-alone NMS entry (tests / reuse)
extern "C" int imcui_hip_simple_nms(imcui_hip_t* h, const float* scores, float* out, int B, int H, int W, int nms_radius,
                                    void* stream_) {
    return nms_launch(h, scores, out, B, H, W, nms_radius, (hipStream_t)stream_);
}
