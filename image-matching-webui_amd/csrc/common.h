// Common device/host helpers for the gfx950 (MI355X) kernels of libimcui_hip.
//
// All contractions use the exact-f32 matrix instruction v_mfma_f32_32x32x2_f32
// (64 FLOP/clk/SIMD, bitwise an fmaf chain) so that results stay within fp32
// round-off of the PyTorch-CPU reference path (keypoint indices / match indices
// must be bit-exact, tensors within 1e-4).
//
// Fragment conventions used everywhere (wave64, lane l, hi = l >> 5, lo = l & 31):
//   A operand : one float per lane = A[row = lo][k = hi]
//   B operand : one float per lane = B[k = hi][col = lo]
//   C/D       : 16 floats per lane, reg r -> row = (r & 3) + 8 * (r >> 2) + 4 * hi, col = lo
// A k-quad (4 consecutive k) is fetched with one 16-byte LDS read; the four MFMA
// steps that consume it pair k = 4*q0 + j (hi = 0 lanes) with k = 4*q1 + j (hi = 1
// lanes) -- any k order is legal as long as A and B agree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define IMCUI_WAVE 64

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- split-precision path: x = hi + lo with hi, lo in f16; a product a*b is evaluated as
// ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_f16 (products exact, f32 accumulate): ~fp32
// accuracy (dropped al*bl term is 2^-22 relative) at 3/16 of the f32-MFMA cost.
// f16 fragment layout (lane l, hi = l >> 5, lo = l & 31): A = 8 halves A[row = lo][k = 8*hi .. 8*hi+7],
// B = 8 halves B[k = 8*hi .. 8*hi+7][col = lo]; C/D as for the f32 instruction.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// hi = round-toward-zero f16 pair (one instruction, saturates instead of overflowing to inf),
// lo = round-to-nearest f16 of the exact remainders
// lo = f16(x - hi) is ONE mixed-precision FMA per element (fma(hi_f16, -1, x_f32) rounded to f16: the
// difference is exact in f32, so this is the same value as subtract-then-convert); the compiler's own
// lowering of the C expression is convert, convert, packed subtract, packed convert.
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    const fp16x2_t h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    hi = __builtin_bit_cast(unsigned, h);
#ifdef IMCUI_SPLIT_NO_ASM
    const f16x2 l2 = {(_Float16)(x0 - (float)h[0]), (_Float16)(x1 - (float)h[1])};
    lo = __builtin_bit_cast(unsigned, l2);
#else
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(x1));
    lo = l;
#endif
}
__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& hi, uint4& lo) {
    split2(a.x, a.y, hi.x, lo.x);
    split2(a.z, a.w, hi.y, lo.y);
    split2(b.x, b.y, hi.z, lo.z);
    split2(b.z, b.w, hi.w, lo.w);
}

// the nearest f16 (round to nearest even, saturating) of 8 floats: the operand of the single-product arithmetic
__device__ __forceinline__ unsigned half2_rtn(float x0, float x1) {
    const f16x2 h = {(_Float16)fminf(fmaxf(x0, -65504.0f), 65504.0f), (_Float16)fminf(fmaxf(x1, -65504.0f), 65504.0f)};
    return __builtin_bit_cast(unsigned, h);
}
// two floats known to lie inside the f16 range -> packed f16, round to nearest even (v_cvt_pk_f16_f32), no clamp
__device__ __forceinline__ unsigned half2_rtn_nc(float x0, float x1) {
    const f16x2 h = {(_Float16)x0, (_Float16)x1};
    return __builtin_bit_cast(unsigned, h);
}
// acc + h.x + h.y for a packed f16 pair: v_dot2_f32_f16 against (1, 1) -- exact products, f32 accumulation
__device__ __forceinline__ float fdot2_ones(unsigned h, float acc) {
    const f16x2 one = {(_Float16)1.0f, (_Float16)1.0f};
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, h), one, acc, false);
}
__device__ __forceinline__ uint4 half8_rtn(const float4& a, const float4& b) {
    return make_uint4(half2_rtn(a.x, a.y), half2_rtn(a.z, a.w), half2_rtn(b.x, b.y), half2_rtn(b.z, b.w));
}

// row inside a 32x32 C/D fragment held by (reg r, half hi)
__device__ __forceinline__ int frag_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// log(sigmoid(x)) = min(x,0) - log1p(exp(-|x|))  (same form ATen uses)
__device__ __forceinline__ float logsigmoidf_(float x) { return fminf(x, 0.0f) - log1pf(expf(-fabsf(x))); }

// GELU(y) = 0.5 y (1 + erf(y / sqrt 2)), erf(t) = 1 - 2^(-t Q(t)) for t = min(|.|, 4): coefficients and the measured
// float32 error (1.3e-7 absolute on erf) from tools/fit/fit_gelu_erf.py.  Branch-free: 7 FMAs and one v_exp_f32.
__device__ __forceinline__ float gelu_poly(float y) {
    const float a = y * 0.70710678118654752440f;
    const float t = fminf(fabsf(a), 4.0f);
    float q = 4.535851622e-05f;
    q = fmaf(q, t, -4.455066228e-04f);
    q = fmaf(q, t, 1.489437302e-03f);
    q = fmaf(q, t, 7.746380288e-04f);
    q = fmaf(q, t, -2.825368941e-02f);
    q = fmaf(q, t, 1.484816223e-01f);
    q = fmaf(q, t, 9.184163809e-01f);
    q = fmaf(q, t, 1.627908587e+00f);
    const float e = __builtin_amdgcn_exp2f(-(t * q));
    const float r = copysignf(1.0f - e, a);
    const float hy = 0.5f * y;
    return fmaf(hy, r, hy);
}

// Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, each with a
// private 4 MiB L2).  Give every XCD a contiguous run of tile ids so that neighbouring tiles
// (which share an operand panel) hit the same L2.  Bijective for any nwg.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// More than 64 KB of dynamic LDS needs `hipFuncAttributeMaxDynamicSharedMemorySize`, and the attribute belongs to the CURRENT device's copy of
// the kernel: a process that drives several GPUs has to opt in on each of them.  `done` = one bit per device ordinal, one word per kernel
// instantiation (a static of the launcher); two threads racing on the same bit both set the attribute, which is harmless.
static inline bool imcui_lds_optin(std::atomic<unsigned long long>& done, const void* kernel, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
}
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// status codes of the C ABI
#define IMCUI_OK 0
#define IMCUI_ERR_ARG -1
#define IMCUI_ERR_WS -2
#define IMCUI_ERR_HIP -3
#define IMCUI_ERR_UNSUPPORTED -4

// A/B switches of the kernel routing.  Read from the environment ONCE, when the handle is created (imcui_hip_create), and changed
// afterwards only through imcui_hip_set_option: the launch paths never call getenv (the plugins run in several UI worker threads, and
// getenv concurrent with a setenv / putenv is a data race in glibc; a switch flipped in the middle of a forward pass would also route
// two launches of one network to different kernels).  Names: the IMCUI_* environment variable without its prefix, lower case.
enum {
    OPT_GEMM_WREG = 0,     // IMCUI_GEMM_WREG: 0 = projections on gemm_split_kernel, 1 = attention-layout projections on gemm_wreg_kernel, 2 (default) = every eligible launch
    OPT_WREG_PIPE,         // IMCUI_WREG_PIPE: 0 = rolled K loop, 1 (default) = three rotating register sets
    OPT_ATTN_VARIANT,      // IMCUI_ATTN_VARIANT: 0 .. 8, default 8, see attention.hip
    OPT_ATTN_SELF,         // IMCUI_ATTN_VARIANT_SELF: -1 (default) = attn_variant; else the variant of LightGlue's SELF blocks in the layers of attn_mix_layers
    OPT_ATTN_CROSS,        // IMCUI_ATTN_VARIANT_CROSS: the same for the CROSS blocks; default -2 = 7 (two-product P.V: audited per block, tools/attn_mix_audit.py) while attn_variant is 8, else attn_variant
    OPT_ATTN_MIX_LAYERS,   // IMCUI_ATTN_MIX_LAYERS: bit l = layer l takes the two overrides above (default 0x1ff: all nine)
    OPT_CONV_TALL,         // IMCUI_CONV_TALL: 0 = 8-row conv tiles everywhere, 1 (default) = 16-row tiles for SuperPoint's fused first layer, 2 = also for plain 64-channel layers
    OPT_CONV_NARROW,       // IMCUI_CONV_NARROW: 0 (default) = 64-output-channel tiles where the 128-channel tiling gives fewer than 256 workgroups; 1 = wherever 128 fit too; 2 = never (A/B)
    OPT_SIMRED,            // IMCUI_SIMRED: 1 (default) = similarities reduced by the persistent kernel of simred.hip, 0 = the round-4 tile GEMM with the reducing epilogue (A/B; mutual-NN only)
    OPT_FFN_TILE,          // IMCUI_FFN_TILE: tokens per workgroup of the fused FFN: 0 (default) = by token count (128 / 64 / 32: the largest that fills the CUs), or 128 / 64 / 32 (bitwise equal results)
    OPT_WREG_TILE,         // IMCUI_WREG_TILE: the same for the weights-in-registers projection GEMM (LightGlue's q / k / v, cross and plain-bias launches)
    OPT_ATTN_SPLIT,        // IMCUI_ATTN_SPLIT: key-split attention launches: 0 never, 1 (default) when the grid has fewer than two workgroups per CU, 2 whenever the caller gave scratch (bitwise equal rows)
    OPT_LOFTR_FINE_SPARSE, // IMCUI_LOFTR_FINE_SPARSE: LoFTR's last FPN stage (the 1/2-resolution fine map): 0 = dense maps, 1 (default) = on the 5x5 windows of the matches when that is cheaper (one 4-byte read-back of the match count), 2 = always on the windows
    OPT_NCNT
};

// optional per-kernel-class HIP-event timing (bench.py's live roofline measurement)
enum { PROF_ATTN = 0, PROF_CONV = 1, PROF_GEMM = 2, PROF_NCLS = 3 };
#define PROF_MAX_EVENTS 4096
struct imcui_hip_s {
    int device;
    int num_cu;
    char err[512];
    int precision;  // 0 = exact f32 MFMA, 1 = 3 x f16 split MFMA (default)
    int prof_on;
    int prof_used[PROF_NCLS];
    int prof_alloc[PROF_NCLS];
    hipEvent_t* prof_ev[PROF_NCLS];  // pairs (start, stop)
    float* lg_dump;  // parity-test hook (imcui_hip_lightglue_set_layer_dump): per-layer token states
    size_t lg_dump_floats;
    long long* ffn_dbg;  // lab hook (imcui_hip_ffn_set_debug): per-workgroup phase stamps of the fused FFN kernel
    // opt-in range check of the split arithmetic (imcui_hip_set_range_check / IMCUI_HIP_CHECK_RANGE=1): device word, bit 0 = an
    // f32 activation beyond the f16 range (|x| > 65504: the hi part saturates, the product is no longer fp32-grade), bit 1 = NaN / Inf
    int* range_flag;
    int opt[OPT_NCNT];  // A/B switches (above)
    int loftr_fine_mode;  // how the last imcui_hip_loftr_forward evaluated the last FPN stage: 0 dense maps, 1 on the windows of the matches (bench.py reports it)
    int loftr_fine_matches;  // the match count that call read back (-1: no read-back)
};
// scan `rows` x `cols` f32 values (row stride ld; rows of sequence s beyond cnt[s] are padding and skipped) into h->range_flag
void imcui_range_check(imcui_hip_s* h, const float* x, long rows, int cols, long ld, const int* cnt, int rows_per_seq, hipStream_t s);
void imcui_prof_begin(imcui_hip_s* h, int cls, hipStream_t s);
void imcui_prof_end(imcui_hip_s* h, int cls, hipStream_t s);

int imcui_set_err(imcui_hip_s* h, int code, const char* fmt, ...);

#define IMCUI_CHECK_LAUNCH(h)                                                          \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess)                                                         \
            return imcui_set_err(h, IMCUI_ERR_HIP, "%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
    } while (0)

// bump allocator over the caller-provided workspace
struct WsAlloc {
    char* base;
    size_t cap;
    size_t off;
    bool ok;
    WsAlloc(void* p, size_t c) : base((char*)p), cap(c), off(0), ok(true) {}
    template <typename T>
    T* get(size_t n) {
        size_t bytes = align_up(n * sizeof(T), 256);
        if (base != nullptr && off + bytes > cap) ok = false;
        T* r = base ? (T*)(base + off) : nullptr;
        off += bytes;
        return r;
    }
};
