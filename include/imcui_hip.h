/* libimcui_hip -- C ABI of the MI355X (gfx950) extract/match backend for imcui.
 *
 * Drop-in boundary.  The reference has no FFI for this path: its hot path
 * is the `_forward(data: dict) -> dict` of three `imcui.hloc` plugins, which delegate to
 * PyTorch modules.  Each entry point below replaces the arithmetic behind one of those calls
 * and is what a host-side binding (ctypes today) binds:
 *
 *   imcui_hip_superpoint_forward  <- imcui/hloc/extractors/superpoint.py:56-57  `self.net(data, self.conf)`
 *   imcui_hip_lightglue_forward   <- imcui/hloc/matchers/lightglue.py:54-75     `self.net(input)`
 *   imcui_hip_mutual_nn           <- imcui/hloc/matchers/nearest_neighbor.py:38-66 `_forward`
 *   imcui_hip_loftr_forward       <- imcui/hloc/matchers/loftr.py:54            `self.net(data_)`
 *   imcui_hip_dual_softmax        <- imcui/hloc/matchers/dual_softmax.py:62-75  `dual_softmax_matcher(...)`
 *   imcui_hip_superglue_forward   <- imcui/hloc/matchers/superglue.py:42-43     `self.net(data)`
 *   *_pack_weights                <- the `_init` weight loading (superpoint.py:48-53, lightglue.py:39-51)
 *
 * Conventions
 *   - every pointer marked [dev] is a device pointer owned by the caller (PyTorch-ROCm tensor
 *     storage); the library never allocates or frees caller-visible memory.  Scratch comes from
 *     a caller-provided workspace sized by the matching *_workspace_bytes() query.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *     All work is enqueued on it; calls return without synchronising unless stated.
 *   - return value: 0 = ok, < 0 = error (IMCUI_HIP_ERR_*); imcui_hip_last_error(h) has the text.
 *     No exceptions cross the boundary.  The library is re-entrant per (handle, stream) and
 *     holds no global mutable state.
 *   - all tensors are float32 / int32, C-contiguous, batch-major.
 */
#ifndef IMCUI_HIP_H
#define IMCUI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMCUI_HIP_OK 0
#define IMCUI_HIP_ERR_ARG -1         /* bad argument */
#define IMCUI_HIP_ERR_WS -2          /* workspace missing / too small */
#define IMCUI_HIP_ERR_HIP -3         /* HIP runtime error */
#define IMCUI_HIP_ERR_UNSUPPORTED -4 /* configuration outside what the kernels support */

typedef struct imcui_hip_s imcui_hip_t;

/* ---- handle ------------------------------------------------------------------------- */
int imcui_hip_create(int device, imcui_hip_t** out);
void imcui_hip_destroy(imcui_hip_t* h);
const char* imcui_hip_last_error(const imcui_hip_t* h);
/* 400 since round 4: imcui_hip_dust3r_forward[_sizes] take the size of the packed buffer (a blob of another layout is rejected),
 * the packed DUSt3R buffer ends in a format trailer, imcui_hip_set_option / _get_option exist.  (100: rounds 1-3.) */
int imcui_hip_version(void);

/* A/B switches of the kernel routing (profiling and the bitwise old-vs-new kernel tests).  The IMCUI_<NAME> environment variables
 * are read ONCE, by imcui_hip_create; afterwards a switch changes only through this call -- set it BETWEEN forward passes, never while
 * another thread runs a call on the same handle.  Names / values: "gemm_wreg" 0 | 1 | 2 (default 2: every eligible projection on
 * the weights-in-registers GEMM), "wreg_pipe" 0 | 1 (default 1), "attn_variant" 8 (default: three f16 products in both contractions; the retired schedules 0 .. 3 and 5 map onto it) | 7 (the two-product P.V; 6 maps
 * onto it; NOT fp32-grade on its own) | 9 (round 6, opt-in: P.V corrections on the block-scaled fp6 matrix instruction, csrc/attention_mx.hip; callers without the
 * fp6 scratch -- SuperGlue, DUSt3R -- fall back to 8), "attn_split" 0 | 1 | 2 (key-split attention launches, csrc/attention.hip: 0 never, 1 = default: when the grid
 * has fewer than two workgroups per CU -- one pair per call --, 2 whenever the caller gave scratch; the geometries are bitwise equal),
 * "attn_variant_self" / "attn_variant_cross" (-1 = attn_variant; else the variant of LightGlue's self / cross blocks in the layers whose bit is set in
 * "attn_mix_layers", default 0x1ff.  Default cross = -2: variant 7 -- the two-product P.V in the CROSS blocks only, audited per block: layer error <= 7.1e-6 and
 * score error <= 4.7e-5 at N = M = 2048 on three weight sets, half the parity bar -- WHILE "attn_variant" is left at 8; an explicit "attn_variant" governs every
 * block; -1 restores three products everywhere), "simred"
 * 1 | 0 (default 1: the mutual-NN matcher on the persistent similarity-and-reduce kernel; 0: the round-4 tile GEMM with the reducing
 * epilogue, kept for A/B and for descriptor widths other than 64 / 128 / 256), "ffn_tile" / "wreg_tile" 0 | 128 | 64 | 32 (tokens per workgroup of the
 * fused FFN / of the weights-in-registers projection GEMM; default 0 = by token count, the largest tile that still gives every CU a workgroup; bitwise equal results), "conv_tall"
 * 0 | 1 | 2 and "conv_narrow" 0 | 1 | 2 (convolution tile shapes; conv_narrow 0 = 64-channel tiles where 128-channel tiles would give fewer than 256
 * workgroups, 1 = always, 2 = never).  Unknown name: IMCUI_HIP_ERR_ARG. */
int imcui_hip_set_option(imcui_hip_t* h, const char* name, int value);
int imcui_hip_get_option(imcui_hip_t* h, const char* name, int* value);

/* Arithmetic mode of the matrix-core kernels (default 1):
 *   0  exact f32: v_mfma_f32_32x32x2_f32, bitwise an fmaf chain
 *   1  3 x f16 split: every f32 operand is split into f16 (hi, lo) and a product is evaluated as
 *      ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_f16 with f32 accumulation (~fp32 accuracy,
 *      ~5x the f32 matrix rate).  Attention then consumes V transposed (internal detail).
 * For imcui_hip_attention_f32 in mode 1 the caller passes pre-split operands: each of Q, K
 * [S][heads][rows][64] and V^T [S][heads][64][rows] as two f16 planes (hi, then lo). */
int imcui_hip_set_precision(imcui_hip_t* h, int mode);
int imcui_hip_get_precision(const imcui_hip_t* h);

/* Optional HIP-event timing of the three heavy kernel classes on the launch stream
 * (0 = attention, 1 = conv3x3, 2 = gemm): enable, run, then read (synchronises the events;
 * returns the summed kernel time in ms and the number of launches, and resets the counters). */
int imcui_hip_profile_enable(imcui_hip_t* h, int on);
int imcui_hip_profile_read(imcui_hip_t* h, int kernel_class, double* total_ms, int* count);

/* ---- SuperPoint --------------------------------------------- */
/* Weight packing runs on the HOST: `w[i]`, `b[i]` are the 12 conv weights (OIHW) / biases in the
 * order conv1a conv1b conv2a conv2b conv3a conv3b conv4a conv4b convPa convPb convDa convDb
 * (upstream state-dict keys `<name>.weight` / `<name>.bias`); `packed` receives
 * imcui_hip_superpoint_packed_floats() floats which the caller uploads once. */
size_t imcui_hip_superpoint_packed_floats(void);
int imcui_hip_superpoint_pack_weights(const float* const* w, const float* const* b, float* packed);

size_t imcui_hip_superpoint_workspace_bytes(int B, int H, int W, int nms_radius);
/* Key-points a non-degenerate HxW image can yield after NMS (sizes `kcap` when max_keypoints < 0). */
int imcui_hip_superpoint_max_keypoints_bound(int H, int W, int nms_radius);

/* image [dev, B,1,H,W] in [0,1]; H, W multiples of 8.  Runtime conf (read per call, like the
 * reference re-reads self.conf): nms_radius (0..4), keypoint_threshold, remove_borders,
 * max_keypoints (-1 = all), fix_sampling (imcui/hloc/extractors/superpoint.py:16-30).
 * Outputs, fixed stride `kcap` per image, first num_keypoints[b] entries valid:
 *   keypoints   [dev, B,kcap,2]   (x, y) float pixel coordinates
 *   scores      [dev, B,kcap]
 *   descriptors [dev, B,kcap,256] row per key-point (the reference's [256,N] is its transpose view)
 *   num_keypoints [dev, B] int32
 *   status      [dev, 1] int32 optional (may be NULL): selection status word of this call, 0 = fine, bit 1 = `kcap`
 *               too small (only possible with max_keypoints = -1 and exactly tied scores; the first kcap are
 *               returned), bit 0 = candidate overflow.  Reading it is the caller's business (no sync here).
 *   score_map   [dev, B,H,W] optional (may be NULL): dense pre-NMS detector scores
 * max_keypoints above 16384 (the on-chip sorter; the UI slider ends at 10000) is rejected with
 * IMCUI_HIP_ERR_UNSUPPORTED before anything is launched; -1 (all) has no such limit.
 * Order: score-descending (ties: lower flat index first) when top-k applies, row-major otherwise. */
int imcui_hip_superpoint_forward(imcui_hip_t* h, const float* packed, const float* image, int B, int H, int W,
                                 int nms_radius, float keypoint_threshold, int remove_borders, int max_keypoints,
                                 int fix_sampling, int kcap, float* keypoints, float* scores, float* descriptors,
                                 int* num_keypoints, int* status, float* score_map, void* ws, size_t ws_bytes, void* stream);
/* Synchronises `stream` and reports selection overflow of the last forward on `ws`
 * (IMCUI_HIP_ERR_UNSUPPORTED + message) -- e.g. kcap too small for a max_keypoints = -1 call. */
int imcui_hip_superpoint_status(imcui_hip_t* h, int B, int H, int W, int nms_radius, void* ws, size_t ws_bytes,
                                void* stream);
/* upstream `simple_nms` alone: scores/out [dev, B,H,W] */
int imcui_hip_simple_nms(imcui_hip_t* h, const float* scores, float* out, int B, int H, int W, int nms_radius,
                         void* stream);

/* ---- LightGlue --------------------------------------------- */
/* Host-side packing of the upstream state dict (9 layers, dim 256, 4 heads).  `tensors` holds the
 * host pointers of imcui_hip_lightglue_num_tensors() tensors; tensor i is the upstream state-dict
 * entry named imcui_hip_lightglue_tensor_name(i) (posenc.Wr.weight, transformers.{l}.*,
 * log_assignment.{l}.*, token_confidence.{l}.token.0.*);
 * `packed` receives imcui_hip_lightglue_packed_floats() floats. */
size_t imcui_hip_lightglue_packed_floats(void);
int imcui_hip_lightglue_num_tensors(void);
const char* imcui_hip_lightglue_tensor_name(int i); /* upstream state-dict key of tensor i */
int imcui_hip_lightglue_pack_weights(const float* const* tensors, float* packed);
/* Variants of upstream's `features` table (imcui/hloc/configs/matchers.py:51-83,140-150: disk- / aliked- / raco- /
 * sift-lightglue).  Call after imcui_hip_lightglue_pack_weights on the same host buffer:
 *   wr [32][wr_cols]: posenc.Wr.weight, wr_cols = 2, or 4 with add_scale_ori (sift, doghardnet: (x, y, scale, orientation));
 *   w_input_proj [256][input_dim], b_input_proj [256]: `input_proj` when input_dim != 256 (128 for disk / aliked / sift);
 *   input_dim: descriptor size, a multiple of 32 and at most 256. */
int imcui_hip_lightglue_pack_input(const float* wr, int wr_cols, const float* w_input_proj, const float* b_input_proj, int input_dim,
                                   float* packed);

size_t imcui_hip_lightglue_workspace_bytes(int B, int ncap);

/* B pairs.  keypoints0/1 [dev, B,ncap,2] pixel (x,y); descriptors0/1 [dev, B,ncap,input_dim] (256 for SuperPoint);
 * scales0/1, oris0/1 [dev, B,ncap] or all NULL: only for weights with add_scale_ori (lightglue.py:62-73 passes them on);
 * n0/n1 [dev, B] int32 valid counts (<= ncap).  size0/size1: (W,H) of the images the key-points
 * live in (only used to normalise key-points, lightglue.py passes `image.shape`).
 * depth_confidence / width_confidence <= 0 disable early stopping / point pruning;
 * pruning_threshold: a side is pruned only while it holds MORE than this many points -- upstream's
 * pruning_keypoint_thresholds[device] (cpu / mps -1 = always, cuda 1024, flash 1536); -1 reproduces the
 * reference's PyTorch-CPU path, which is what the parity tests pin;
 * filter_threshold is conf["match_threshold"] (imcui/hloc/matchers/lightglue.py:50).  The three
 * thresholds are doubles because the reference compares fp32 tensors against Python floats
 * (e.g. `1 - width_confidence` is formed in double before the fp32 cast).
 * Outputs [dev]: matches0/1 [B,ncap] int32 (-1 = unmatched), matching_scores0/1 [B,ncap],
 * stop [B] int32 (layers run), prune0/1 [B,ncap] int32.  Entries >= n are -1 / 0. */
int imcui_hip_lightglue_forward(imcui_hip_t* h, const float* packed, int B, int ncap, int input_dim, const float* keypoints0,
                                const float* keypoints1, const float* descriptors0, const float* descriptors1,
                                const float* scales0, const float* oris0, const float* scales1, const float* oris1,
                                const int* n0, const int* n1, float w0, float h0, float w1, float h1,
                                double depth_confidence, double width_confidence, int pruning_threshold,
                                double filter_threshold, int* matches0, int* matches1, float* matching_scores0,
                                float* matching_scores1, int* stop, int* prune0, int* prune1, void* ws, size_t ws_bytes,
                                void* stream);
/* Parity-test hook (the oracle exposes the same intermediates): while `dump` [dev] is non-NULL every
 * imcui_hip_lightglue_forward on this handle copies the token states x [2B, R, 256] (R = roundup(ncap, 128),
 * row (2*pair + image)*R + i, rows in their current -- pruned -- order) after each of the 9 layers to
 * dump + layer * 2B*R*256; `floats` = capacity of `dump`.  NULL switches it off. */
int imcui_hip_lightglue_set_layer_dump(imcui_hip_t* h, float* dump, size_t floats);

/* ---- SuperGlue (imcui/hloc/matchers/superglue.py:42-43 `self.net(data)`) ---------- */
/* Host-side packing of the upstream state dict (Vincentqyw/SuperGluePretrainedNetwork models/superglue.py:
 * kenc.encoder.*, gnn.layers.{0..17}.{attn.proj.{0,1,2},attn.merge,mlp.{0,1,3}}, final_proj, bin_score).
 * `tensors` holds the host pointers of imcui_hip_superglue_num_tensors() tensors, tensor i being the state-dict entry
 * imcui_hip_superglue_tensor_name(i) (Conv1d weights [out,in,1] are passed as [out,in]).  Eval-mode BatchNorm and
 * attn.merge are folded into the neighbouring layers; heads are de-interleaved. */
size_t imcui_hip_superglue_packed_floats(void);
int imcui_hip_superglue_num_tensors(void);
const char* imcui_hip_superglue_tensor_name(int i);
int imcui_hip_superglue_pack_weights(const float* const* tensors, float* packed);
size_t imcui_hip_superglue_workspace_bytes(int B, int ncap);
/* B pairs.  keypoints0/1 [dev, B,ncap,2] pixel (x,y); scores0/1 [dev, B,ncap] detector scores; descriptors0/1
 * [dev, B,ncap,256] (row per key-point); n0/n1 [dev, B] int32 valid counts (<= ncap); (w0,h0)/(w1,h1): size of the
 * images (superglue.py passes `data["image0"].shape`, only used to normalise key-points).
 * sinkhorn_iterations / match_threshold = conf (superglue.py:17-18; configs/matchers.py:15-16,30-31).
 * Outputs [dev]: matches0/1 [B,ncap] int32 (-1 = unmatched), matching_scores0/1 [B,ncap]; a pair with an empty
 * image gets all -1 / 0 (upstream early return).  Entries >= n are -1 / 0. */
int imcui_hip_superglue_forward(imcui_hip_t* h, const float* packed, int B, int ncap, const float* keypoints0,
                                const float* keypoints1, const float* scores0, const float* scores1, const float* descriptors0,
                                const float* descriptors1, const int* n0, const int* n1, float w0, float h0, float w1, float h1,
                                int sinkhorn_iterations, double match_threshold, int* matches0, int* matches1,
                                float* matching_scores0, float* matching_scores1, void* ws, size_t ws_bytes, void* stream);

/* ---- LoFTR (kornia.feature.LoFTR behind imcui/hloc/matchers/loftr.py:54) ---- */
/* Host-side packing.  Every convolution / Linear is one layer W[N][K] (+ bias[N], may be NULL) in GEMM
 * layout: imcui_hip_loftr_num_layers() layers of imcui_hip_loftr_layer_shape(i); convolutions as
 * [Cout][tap][Cin] with BatchNorm folded and the 196-channel stage zero-padded to 256 channels (the
 * Python host layer imcui_hip/backend.py:pack_loftr builds them from the kornia state dict).
 * conv1_w [49][128] / conv1_b [128]: the first 7x7 convolution (BN folded).  norms: LayerNorm weight /
 * bias vectors, imcui_hip_loftr_num_norms() of imcui_hip_loftr_norm_dim(i) floats. */
size_t imcui_hip_loftr_packed_floats(void);
int imcui_hip_loftr_num_layers(void);
int imcui_hip_loftr_layer_shape(int i, int* N, int* K);
int imcui_hip_loftr_num_norms(void);
int imcui_hip_loftr_norm_dim(int i);
int imcui_hip_loftr_pack_weights(const float* conv1_w, const float* conv1_b, const float* const* w, const float* const* b,
                                 const float* const* norms, float* packed);
size_t imcui_hip_loftr_workspace_bytes(int B, int H0, int W0, int H1, int W1);
/* kornia LoFTR.forward on B pairs: image0 [dev, B,1,H0,W0], image1 [dev, B,1,H1,W1] (multiples of 8, >= 32; the two
 * sizes may differ -- `minima_loftr` keeps each image's aspect ratio, configs/matchers.py:283 -- kornia then runs the
 * backbone per image instead of on the concatenated batch, which gives the same values).
 * Outputs with capacity B*(H0/8)*(W0/8) rows, first num_matches[0] valid, ordered like torch.where
 * (batch-major, coarse cell of image0 ascending): keypoints0/1 [dev, cap,2] pixel (x,y),
 * confidence [dev, cap], batch_indexes [dev, cap] int32, num_matches [dev, 1] int32.
 * match_threshold = conf["match_threshold"] (loftr.py:23); temp_bug_fix = 1 only for MINIMA weights (:28). */
int imcui_hip_loftr_forward(imcui_hip_t* h, const float* packed, const float* image0, const float* image1, int B, int H0,
                            int W0, int H1, int W1, double match_threshold, int temp_bug_fix, float* keypoints0,
                            float* keypoints1, float* confidence, int* batch_indexes, int* num_matches, void* ws,
                            size_t ws_bytes, void* stream);

/* The last FPN stage (1/2 resolution: layer1_outconv + layer1_outconv2, kornia ResNetFPN_8_2 behind loftr.py:54) produces the fine map that is
 * only ever read through the 5x5 windows of the matched cells.  Option "loftr_fine_sparse" (imcui_hip_set_option; default 1): the stage is
 * deferred until the matches are known and evaluated on those windows alone when that is cheaper than the dense maps -- for that decision
 * imcui_hip_loftr_forward reads the 4-byte match count back and SYNCHRONISES THE STREAM ONCE per call (not while the stream is being captured
 * into a graph: dense then); 2 = always on the windows, 0 = dense maps, no read-back (the call does not synchronise).  Same matches either way;
 * window features agree to round-off (tests/test_gpu_loftr.py).  Returns how the last forward on this handle did it: 0 dense, 1 windows;
 * *matches (may be NULL) = the count it read back, -1 without a read-back. */
int imcui_hip_loftr_last_fine_mode(imcui_hip_t* h, int* matches);

/* byte offset inside the LoFTR workspace of: 0 coarse features after the transformer [B*L0 + B*L1, 256] (side 0 first),
 * 1 fine features [B*H0/2*W0/2 + B*H1/2*W1/2, 128], 2 sim [B,L0,L1], 3 fine windows [2,B*L0,25,128]  (parity tests) */
size_t imcui_hip_loftr_debug_offset(int which, int B, int H0, int W0, int H1, int W1);

/* ---- EfficientLoFTR (upstream zju3dv/EfficientLoFTR `LoFTR` behind
 * imcui/hloc/matchers/eloftr.py:79, 'full' model, fp32) ------------------------------------------------------ */
/* Host-side packing, as for LoFTR: imcui_hip_eloftr_num_layers() layers W[N][K] (+ bias, may be NULL) of
 * imcui_hip_eloftr_layer_shape(i) in GEMM layout -- 20 re-parameterised RepVGG blocks ([Cout][tap][Cin]; 3x3 + 1x1 +
 * identity branches and their BatchNorms folded, what the reference's `reparameter()` does, eloftr.py:61), 8 attention
 * blocks x {q, k, v, o, fc1, fc2}, 7 fine-fusion convolutions (BatchNorm folded, the 1/16 coarse-feature scale folded
 * into the first) -- built from the state dict by imcui_hip/backend.py:pack_eloftr.  conv0_w [9][64] / conv0_b [64]: the
 * first block (1 -> 64, stride 2).  dw: 8 depth-wise 4x4 query-aggregation kernels [256][16] (block = layer * 2 +
 * {self, cross}).  norms: 32 vectors of 256 (per block: aggregation norm w, b; mlp LayerNorm w, b).  inv_freq [64]:
 * rotary frequencies 1 / 10000^(2t/128). */
size_t imcui_hip_eloftr_packed_floats(void);
int imcui_hip_eloftr_num_layers(void);
int imcui_hip_eloftr_layer_shape(int i, int* N, int* K);
int imcui_hip_eloftr_pack_weights(const float* conv0_w, const float* conv0_b, const float* const* w, const float* const* b,
                                  const float* const* dw, const float* const* norms, const float* inv_freq, float* packed);
size_t imcui_hip_eloftr_workspace_bytes(int B, int H0, int W0, int H1, int W1, int debug_windows);
/* Upstream LoFTR.forward on B pairs: image0 [dev, B,1,H0,W0], image1 [dev, B,1,H1,W1] (multiples of 32, >= 64; the UI path
 * resizes both images to width x height, configs/matchers.py:296-303, the batch path of match_dense.py does not, so the two
 * sizes may differ: the backbone then runs side by side and the attention is between token sets of different length).
 * Outputs with capacity B*(H0/8)*(W0/8) rows, first num_matches[0] valid, ordered batch-major by the coarse cell of image0: keypoints0/1
 * [dev, cap,2] pixel (x,y) after the two-stage fine refinement, confidence [dev, cap], batch_indexes [dev, cap] int32,
 * num_matches [dev, 1] int32.  match_threshold = conf["match_threshold"] (eloftr.py:54).  debug_windows != 0 also
 * writes the unfolded fine windows to the workspace (parity tests; size it with the same flag). */
int imcui_hip_eloftr_forward(imcui_hip_t* h, const float* packed, const float* image0, const float* image1, int B, int H0, int W0,
                             int H1, int W1, double match_threshold, float* keypoints0, float* keypoints1, float* confidence,
                             int* batch_indexes, int* num_matches, int debug_windows, void* ws, size_t ws_bytes, void* stream);
/* The same with the arithmetic of the reference wrapper's `precision` switch (imcui/hloc/matchers/eloftr.py:32-33,43-47,63-64):
 * arith 0 = the call above ("fp32": 3 x f16 split products, fp32-grade); arith 1 = "fp16" / "mp": one f16 product per element pair
 * (f32 accumulate) in the backbone and fine-fusion convolutions, everything that decides a match in the split arithmetic. */
int imcui_hip_eloftr_forward_ex(imcui_hip_t* h, const float* packed, const float* image0, const float* image1, int B, int H0, int W0,
                                int H1, int W1, double match_threshold, int arith, float* keypoints0, float* keypoints1,
                                float* confidence, int* batch_indexes, int* num_matches, int debug_windows, void* ws, size_t ws_bytes,
                                void* stream);
/* byte offset inside the workspace of (per-image buffers: the B maps of image 0, then the B maps of image 1): 0 backbone 1/2
 * features [.,H/2,W/2,64], 1 1/4 features [.,H/4,W/4,128], 2 coarse features after the transformer [.,L,256], 3 sim [B,L0,L1],
 * 4 fused 1/2-resolution fine map [.,H/2,W/2,64], 5 fine windows [B*L0][64 + 100][64] (debug_windows)  (parity tests) */
size_t imcui_hip_eloftr_debug_offset(int which, int B, int H0, int W0, int H1, int W1);

/* ---- nearest neighbour by dot product (the primitive of MASt3R's `fast_reciprocal_NNs(..., dist="dot")`,
 * imcui/hloc/matchers/mast3r.py:68-75 -> upstream mast3r/fast_nn.py `cdistMatcher.query`) --------------------------------- */
/* idx [dev, Q] int32 = FIRST arg-max over n of <queries[q], db[n]>, best [dev, Q] (may be NULL) the maximum; queries [dev, Q,D], db [dev,
 * N,D] f32 rows, D in {16, 24, 32}.  Exact-f32 MFMA, similarity and running arg-max fused (no Q x N matrix). */
size_t imcui_hip_nn_argmax_workspace_bytes(int Q, int N);
int imcui_hip_nn_argmax_f32(imcui_hip_t* h, const float* queries, const float* db, int Q, int N, int D, int* idx, float* best, void* ws,
                            size_t ws_bytes, void* stream);
/* The same search in the 3 x f16 split arithmetic (rows scaled by 2^8, split into f16 hi / lo planes, three products per pair,
 * f32 accumulate; D <= 32): a quarter of the matrix cycles.  fp32-grade values, but not the exact-f32 instruction's fmaf chain:
 * between candidates closer than ~3e-7 the arg-max may differ from imcui_hip_nn_argmax_f32's (opt-in, conf "matcher_arithmetic"). */
size_t imcui_hip_nn_argmax_split_workspace_bytes(int Q, int N);
int imcui_hip_nn_argmax_split_f32(imcui_hip_t* h, const float* queries, const float* db, int Q, int N, int D, int* idx, float* best, void* ws,
                                  size_t ws_bytes, void* stream);

/* ---- DUSt3R pair network (BASELINE config 5; what `dust3r.inference.inference(pairs, self.net, ...)`
 * computes at imcui/hloc/matchers/duster.py:73 with self.net = AsymmetricCroCo3DStereo (duster.py:37): ViT encoder with 2-D rotary
 * embedding, two-stream cross-attention decoder, DPT point-map head; the un-vendored `third_party/dust3r` submodule) -------- */
/* The architecture is given by (enc_dim, enc_depth, dec_dim, dec_depth): 1024, 24, 768, 12 for `duster_vit_large.pth`; dims are
 * multiples of 64 (head_dim 64) up to 1024, dec_depth a multiple of 4 (DPT hooks at 0, dec_depth/2, 3 dec_depth/4, dec_depth).
 * Host-side packing: imcui_hip_dust3r_num_layers() matrices W[N][K] of imcui_hip_dust3r_layer_shape(i) (+ bias [N] or NULL):
 * patch_embed.proj as [E][3*16*16]; per encoder block qkv, proj, fc1, fc2; decoder_embed; per decoder block of dec_blocks then
 * dec_blocks2: qkv, proj, cross projq, [projk ; projv], cross proj, fc1, fc2; per head (downstream_head1, 2): act_postprocess
 * 1x1 convolutions, the kernel = stride transposed convolutions as [(dy, dx, cout)][cin] with the bias tiled, the 3x3
 * convolutions in implicit-GEMM order [cout][tap][cin] (act_postprocess.3.1, layer_rn, refinenet 4..1 residual units + out_conv,
 * head.0, head.2).  imcui_hip_dust3r_num_vectors() f32 vectors of imcui_hip_dust3r_vector_len(i): the LayerNorm weights / biases in
 * module order, head.4 weight [4][128] and bias [4] per head, the 16 rotary frequencies 100^(-i/16).  Built from the upstream
 * state dict by imcui_hip/backend.py:pack_dust3r.  CONTRACT since round 3: the q / k / v projections (attn.qkv, cross_attn.projq,
 * [projk ; projv]) and mlp.fc1 are handed over with the affine part of the LayerNorm in front of them folded in -- W * gamma (per input
 * column) and b + W beta for norm1 / norm2 / norm_y / norm3 -- because the device normalises those inputs WITHOUT gamma / beta (one
 * pass serves several consumers); their LayerNorm vectors are still passed (layout unchanged) and ignored.  enc_norm and dec_norm
 * are applied as they are. */
/* desc_dim: 0 = DUSt3R; > 0 = MASt3R (`AsymmetricMASt3R`, imcui/hloc/matchers/mast3r.py:41 with the 'catmlp+dpt' head of
 * `MASt3R_ViTLarge_BaseDecoder_512_catmlpdpt_metric.pth`, desc_dim = 24): two more matrices per head, head_local_features.fc1
 * [4 (E + D)][E + D] and .fc2 [(desc_dim + 1) * 256][4 (E + D)], after head.2. */
size_t imcui_hip_dust3r_packed_floats(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim);
int imcui_hip_dust3r_num_layers(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim);
int imcui_hip_dust3r_num_vectors(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim);
int imcui_hip_dust3r_layer_shape(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, int i, int* N, int* K);
int imcui_hip_dust3r_vector_len(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, int i);
/* float offsets of layer i inside the packed buffer (inspection / tests): bias, f16 hi / lo planes, 2^-e scale; kind 0 = fragment-major
 * GEMM planes, 1 = 3x3 convolution planes */
int imcui_hip_dust3r_layer_offsets(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, int i, size_t* bias, size_t* plane_hi,
                                   size_t* plane_lo, size_t* scale, int* kind);
int imcui_hip_dust3r_pack_weights(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, const float* const* w,
                                  const float* const* b, const float* const* vec, float* packed);
/* Format of the packed buffer (round 4).  Its CONTENT is a contract with the packer: the LayerNorm gamma / beta of every block must
 * already be folded into the matrices that consume the normalised rows (qkv, the cross-attention projections, fc1: what
 * imcui_hip/backend.py:dust3r_matrices does before it calls imcui_hip_dust3r_pack_weights) -- the device only normalises -- and the
 * row sums of the folded matrices sit behind the vectors.  The buffer ends in a 64-word trailer (magic 'IMDU', format number, total
 * size, the five configuration integers).  `packed_floats` of the forward entry points = the size of the caller's buffer: anything but
 * imcui_hip_dust3r_packed_floats() of the configuration is refused with IMCUI_HIP_ERR_ARG (a blob cached from an older library would
 * otherwise run with its affine parts dropped and return plausible but wrong point maps); imcui_hip_dust3r_check_packed verifies
 * size AND trailer of a HOST copy (0 = fine) -- call it once when a cached blob is loaded. */
int imcui_hip_dust3r_format_version(void);
int imcui_hip_dust3r_check_packed(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, const float* packed_host, size_t packed_floats);
size_t imcui_hip_dust3r_workspace_bytes(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, int NI, int P, int H, int W);
size_t imcui_hip_dust3r_dump_floats(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, int NI, int P, int H, int W);
/* `AsymmetricCroCo3DStereo.forward(view1, view2)` for P directed pairs over NI images: images [dev, NI,3,H,W] in [0,1] (the
 * wrapper's mean = std = 0.5 normalisation, duster.py:60-64, is applied inside), H and W multiples of 16 (the patch size); pairs [dev, P,2] int32 =
 * (view-1 image, view-2 image) -- duster.py:70-72 asks for (1,0) then (0,1) (`make_pairs`' order).  Every image is encoded once.  Outputs, view-major:
 * pts3d [dev, 2,P,H,W,3] (view 1: `pts3d`; view 2: `pts3d_in_other_view`, i.e. in view 1's frame), conf [dev, 2,P,H,W]; with
 * desc_dim > 0 also desc [dev, 2,P,H,W,desc_dim] (unit-norm local descriptors: MLP on [encoder | decoder] tokens, pixel shuffle
 * 16, `desc / |desc|`) and desc_conf [dev, 2,P,H,W] = exp(.) (mast3r.py:61-64 reads `pred1["desc"]`, `pred2["desc"]`); NULL otherwise.
 * dump (may be NULL; parity tests): imcui_hip_dust3r_dump_floats() floats of intermediate token states and head maps.
 * arith: 0 = 3 x f16 split products (fp32-grade results, the parity mode), 1 = one f16 product per element pair in the GEMMs and
 * convolutions with f32 accumulation (the class of the bf16 run the reference's configuration names; since round 3 attention too: hi planes of Q / K / V, P rounded to f16).
 * The handle must be in the split mode (IMCUI_ERR_UNSUPPORTED in precision 0). */
int imcui_hip_dust3r_forward(imcui_hip_t* h, int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, const float* packed,
                             size_t packed_floats, const float* images, int NI, int H, int W, const int* pairs, int P, int arith, float* pts3d, float* conf,
                             float* desc, float* desc_conf, float* dump, size_t dump_floats, void* ws, size_t ws_bytes, void* stream);

/* The same network on images of SEVERAL sizes.  imcui/hloc/match_dense.py:match_images and ImagePairDataset.preprocess resize
 * each image of a pair on its own (resize_max 512, dfactor 16), so two photos of different aspect ratio reach duster.py:73 at two
 * sizes; upstream's `inference` then encodes the two views separately and runs one pair per batch (dust3r/inference.py).
 * sizes [host, NI,2] = (H, W) of every image, multiples of 16, at most 4 distinct sizes per call; images [dev]: the NI images
 * [3,H_i,W_i] one behind the other; pairs_host [host, P,2] and pairs [dev, P,2]: the same table twice (the host copy lays out the
 * sequences, the device copy drives the gathers).  Outputs are ragged, map after map in (view, pair) order: the map of
 * (view v, pair p) has the size of image pairs[p][v] and starts at pixel map_pixel_offsets[v P + p] -- pts3d at 3 x that offset,
 * conf at it, desc at desc_dim x it; map_pixel_offsets [host, 2 P + 1] (may be NULL) is filled by the call, the last entry is the
 * total number of pixels the output buffers must hold.  With one size this is exactly imcui_hip_dust3r_forward's layout.
 * dump (may be NULL): imcui_hip_dust3r_token_dump_floats() floats -- the token states only ((enc_depth + 2) x [NI,R,E], (dec_depth + 2) x
 * [2P,R,D], R = tokens of the largest image rounded up to 128; rows past a sequence's own token count are undefined). */
size_t imcui_hip_dust3r_workspace_bytes_sizes(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, int NI, const int* sizes, int P);
size_t imcui_hip_dust3r_token_dump_floats(int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, int NI, const int* sizes, int P);
int imcui_hip_dust3r_forward_sizes(imcui_hip_t* h, int enc_dim, int enc_depth, int dec_dim, int dec_depth, int desc_dim, const float* packed,
                                   size_t packed_floats, const float* images, int NI, const int* sizes, const int* pairs_host, const int* pairs, int P, int arith,
                                   float* pts3d, float* conf, float* desc, float* desc_conf, size_t* map_pixel_offsets, float* dump,
                                   size_t dump_floats, void* ws, size_t ws_bytes, void* stream);

/* ---- baseline-JPEG decode (the step before the path) ----------------------------------------------
 * imcui/hloc/utils/io.py:11-21 `read_image` = cv2.imread(IMREAD_GRAYSCALE | IMREAD_COLOR), called per image by
 * imcui/hloc/extract_features.py:120-156 and match_dense.py.  Split where the format splits (csrc/jpeg.hip): the Huffman bit stream is
 * decoded on HOST threads (imcui_hip_jpeg_info / _entropy_decode: re-entrant, no state, no handle), dequantisation + 8x8 inverse DCT +
 * chroma up-sampling + YCbCr -> RGB run on the DEVICE (imcui_hip_jpeg_reconstruct).  The device arithmetic restates libjpeg's default
 * path (jpeg_idct_islow, fancy up-sampling, ycc_rgb_convert) integer for integer: outputs equal PIL's / cv2's decode BIT FOR BIT
 * (oracle/jpeg.py is pinned to PIL on every JPEG of the reference repository).
 * info [host, 24 ints]: 0 width, 1 height, 2 components (1 | 3), 3 hmax, 4 vmax, 5 MCUs per row, 6 MCU rows, 7 restart interval,
 * 8 EXIF orientation (1 = upright or absent; 2..8: imcui_hip_orient_u8 after the reconstruction), 9 + 4c .. 11 + 4c: h, v, quantisation
 * table of component c.  Supported: SOF0 / SOF1 Huffman, 8 bit, 1 or 3 components, 4:4:4 / 4:2:2 / 4:2:0, restart intervals,
 * interleaved or per-component scans, and -- round 5 -- progressive Huffman frames (SOF2: every scan up to EOI is accumulated into the same
 * coefficient planes); everything else (arithmetic coding, CMYK, 12 bit, 4:4:0) returns IMCUI_HIP_ERR_UNSUPPORTED. */
int imcui_hip_jpeg_info(const unsigned char* data, size_t n, int* info);
/* number of int16 coefficients of all components (every component padded to whole MCUs) */
size_t imcui_hip_jpeg_coef_count(const int* info);
/* coef [host, imcui_hip_jpeg_coef_count() int16]: quantised DCT coefficients in natural order, [component][block row][block column][64];
 * qt [host, 3 x 64 uint16]: the quantisation table of every component, natural order */
int imcui_hip_jpeg_entropy_decode(const unsigned char* data, size_t n, short* coef, unsigned short* qt);
/* `count` files on `threads` host threads OF THE LIBRARY (no interpreter lock, no per-file allocation): planes [3 * count] = where
 * component c of file i goes (planes[3 i + c]; unused components may be NULL) -- typically slices of one pinned staging buffer laid out
 * plane-major, so the luma coefficients of a batch cross PCIe in one transfer; qt [count][3 * 64]; status [count] per-file codes */
int imcui_hip_jpeg_entropy_decode_batch(const unsigned char* const* data, const size_t* sizes, int count, short* const* planes, unsigned short* qt,
                                        int* status, int threads);
size_t imcui_hip_jpeg_workspace_bytes(const int* info, int gray);
size_t imcui_hip_jpeg_workspace_bytes_batch(const int* info, int gray, int n);
/* n files of ONE geometry (equal info records) in three launches: coef_y / coef_cb / coef_cr [dev, n x plane coefficients] (chroma may be
 * NULL for gray output / one-component files), qt [dev, n x 192], out [dev]: [n,H,W] uint8 (gray) or [n,H,W,3] */
/* EXIF orientation 1..8 applied to a decoded image (cv2.imread does this inside the decoder): src [dev, H,W,C] uint8 -> dst [dev, H*W*C
 * bytes] = [H,W,C] for 1..4, [W,H,C] for 5..8 (5 transpose, 6 rotate 90 clockwise, 7 transverse, 8 rotate 90 counter-clockwise) */
int imcui_hip_orient_u8(imcui_hip_t* h, const unsigned char* src, int H, int W, int C, int orientation, unsigned char* dst, void* stream);
int imcui_hip_jpeg_reconstruct_batch(imcui_hip_t* h, const short* coef_y, const short* coef_cb, const short* coef_cr, const unsigned short* qt, const int* info,
                                     int n, int gray, unsigned char* out, void* ws, size_t ws_bytes, void* stream);
/* coef / qt [dev]: copies of the two buffers above; info [host]; out [dev]: gray != 0 -> [H][W] uint8 = the luma plane (what
 * cv2.IMREAD_GRAYSCALE returns for a YCbCr file), else [H][W][3] RGB (a one-component file replicated, as IMREAD_COLOR does) */
int imcui_hip_jpeg_reconstruct(imcui_hip_t* h, const short* coef, const unsigned short* qt, const int* info, int gray, unsigned char* out,
                               void* ws, size_t ws_bytes, void* stream);

/* ---- batched geometric verification (the step after the path) ------------------------------------------------
 * imcui/ui/utils.py:424-456 `proc_ransac_matches` (cv2.findHomography / findFundamentalMat per pair on the host, called twice per pair by
 * `compute_geometry` :532-610).  An ADDITIONAL method for the reference's `ransac_zoo` ("HIP_RANSAC"), not a re-implementation of cv2's
 * USAC samplers: plain RANSAC with local optimisation for B pairs at once, every step specified in csrc/geometry.hip and restated on the
 * CPU in oracle/geometry.py (parity unpinned with respect to cv2).
 * pts0 / pts1 [dev, B,N,2] float32 matched key-points in pixels (row i of pair b is one correspondence), counts [dev, B] valid rows;
 * geometry 0 = homography (x1 ~ H x0, forward transfer error), 1 = fundamental matrix (x1^T F x0 = 0, Sampson error);
 * reproj_threshold [px], confidence, max_iter as in the reference's call (at most 16384 hypotheses are drawn), seed: the sampler is a
 * counter-based generator, results are a pure function of (inputs, seed).
 * model [dev, B,9] float64 row-major (H with h33 = 1; F with unit Frobenius norm), mask [dev, B,N] uint8 inliers, info [dev, B,4] int32 =
 * (inliers, hypotheses consumed by the sequential stopping rule, index of the winning hypothesis, ok). */
size_t imcui_hip_ransac_workspace_bytes(int B, int N, int max_iter);
int imcui_hip_ransac(imcui_hip_t* h, const float* pts0, const float* pts1, const int* counts, int B, int N, int geometry, double reproj_threshold,
                     double confidence, int max_iter, unsigned long long seed, double* model, unsigned char* mask, int* info, void* ws, size_t ws_bytes,
                     void* stream);

/* ---- mutual nearest neighbour (row a12) --------------------------------------------------- */
/* Round 5: descriptors of width 64 / 128 / 256 (SIFT, DISK, SuperPoint ...) run on the persistent similarity-and-reduce kernel
 * (csrc/simred.hip) in EITHER arithmetic: both sets are packed once into matrix-core fragments (4 bytes per element), a workgroup keeps
 * 128 descriptors of image 0 in registers and streams image 1 past them; per row (best, index, second best) live in registers, per
 * column and 128-row block 12 bytes go to memory.  The similarity matrix never exists.  Other widths keep the round-4 paths: the tile
 * GEMM with the reducing epilogue (3 x f16 split) or the materialised B x N x M matrix (exact f32: 4 B N M bytes).
 * Workspace: ..._bytes(B, N, M) serves every path (sized for D <= 256); ..._bytes_for(h, ...) what THIS handle needs for D <= 256;
 * ..._bytes_d(h, ..., D) what this handle needs for descriptors of width D. */
size_t imcui_hip_mutual_nn_workspace_bytes(int B, int N, int M);
size_t imcui_hip_mutual_nn_workspace_bytes_for(imcui_hip_t* h, int B, int N, int M);
size_t imcui_hip_mutual_nn_workspace_bytes_d(imcui_hip_t* h, int B, int N, int M, int D);
/* desc0 [dev, B,N,D], desc1 [dev, B,M,D] row per descriptor (D % 32 == 0); ratio_threshold /
 * distance_threshold <= 0 mean "None"; matches0 [dev, B,N] int32, scores0 [dev, B,N]. */
int imcui_hip_mutual_nn(imcui_hip_t* h, const float* desc0, const float* desc1, int B, int N, int M, int D,
                        double ratio_threshold, double distance_threshold, int do_mutual_check, int* matches0,
                        float* scores0, void* ws, size_t ws_bytes, void* stream);
/* The same matcher on the layout `NearestNeighbor._forward` receives (nearest_neighbor.py:38-66): descriptors0 [dev, B,D,N], descriptors1
 * [dev, B,D,M], one column per descriptor.  D = 64 / 128 / 256: the fragment packer reads that layout directly (no transposed copy);
 * other widths: transposed on the device into the workspace (tiled through LDS), then the call above. */
size_t imcui_hip_mutual_nn_dn_workspace_bytes_for(imcui_hip_t* h, int B, int N, int M, int D);
int imcui_hip_mutual_nn_dn(imcui_hip_t* h, const float* desc0_dn, const float* desc1_dm, int B, int N, int M, int D, double ratio_threshold,
                           double distance_threshold, int do_mutual_check, int* matches0, float* scores0, void* ws, size_t ws_bytes, void* stream);

/* ---- PNG decode (row f-3; `read_image` = cv2.imread, imcui/hloc/utils/io.py:11-21; the reference's WxBS / EVD fixtures are PNG) ----
 * Host: chunk walk + zlib inflate of the IDAT stream into the FILTERED scan lines (re-entrant; the batch call runs on the library's own host
 * threads into caller-provided, typically pinned, staging).  Device: the scan-line filters undone as a wavefront (one workgroup per image,
 * one thread per row) and the pixels written as the host reader returns them: [H][W] for gray / gray + alpha files, [H][W][3] RGB for RGB /
 * RGBA / palette files (alpha dropped).  Bit-exact (checker: PIL).  8-bit, non-interlaced files; everything else: IMCUI_HIP_ERR_UNSUPPORTED
 * (the caller keeps its host reader).  info: 8 ints = width, height, colour type, samples per pixel, decoded channels (1 | 3), bytes per
 * pixel, palette entries, 0. */
int imcui_hip_png_info(const unsigned char* data, size_t n, int* info);
size_t imcui_hip_png_raw_bytes(const int* info);
int imcui_hip_png_inflate(const unsigned char* data, size_t n, unsigned char* raw, size_t raw_bytes, unsigned char* palette);
int imcui_hip_png_inflate_batch(const unsigned char* const* data, const size_t* sizes, int count, unsigned char* const* raw, const size_t* raw_bytes,
                                unsigned char* palettes, int* status, int threads);
size_t imcui_hip_png_workspace_bytes(const int* infos, int count);
/* raw [dev] + raw_offsets [host]: scan lines of file i at raw + raw_offsets[i]; infos [host, count x 8]; palettes [dev, count x 768] or NULL;
 * out [dev] + out_offsets [host]: the decoded image of file i. */
int imcui_hip_png_reconstruct_batch(imcui_hip_t* h, const unsigned char* raw, const size_t* raw_offsets, const int* infos, const unsigned char* palettes, int count,
                                    unsigned char* out, const size_t* out_offsets, void* ws, size_t ws_bytes, void* stream);

/* ---- test hook: ONE launch of the similarity-and-reduce kernel (csrc/simred.hip) on raw matrices (tests/test_gpu_simred.py) ----
 * sim[b] = alpha * A[b] . Bm[b]^T is reduced, never stored.  A [dev, batch,M,K], Bm [dev, batch,N,K] f32, K in {64, 128, 256}; mcnt / ncnt
 * [dev, batch] live rows / columns per batch or NULL.  mode 0: nearest neighbours (best, first index, second best); 1: soft-max statistics
 * (max, sum exp); 2: dual-softmax confidence, row best + first column, column best (needs rmax / rsum / cmax / csum, optional tile flags
 * [batch][ceil(M/128)][ceil(N/128)]); 3: LightGlue's log assignment, row best + first column, column best + first row (rsum / csum = LOG
 * sums, l0 [batch,M], l1 [batch,N]); 4: mode 0 without the second best (r1 / c1 untouched: find_nn without a ratio test).  alpha > 0, and 1
 * for the nearest-neighbour modes.  Row outputs r0 / r1 / ri [batch][nchunk][M] (nchunk 0: imcui_hip_simred_chunks), column outputs
 * c0 / c1 / ci [batch][ceil(M/128)][N]. */
int imcui_hip_simred_chunks(int batch, int M, int N);
size_t imcui_hip_simred_debug_workspace_bytes(int batch, int M, int N, int K);
int imcui_hip_simred_debug(imcui_hip_t* h, int mode, const float* A, const float* Bm, int batch, int M, int N, int K, const int* mcnt, const int* ncnt, float alpha,
                           int nchunk, float* r0, float* r1, int* ri, float* c0, float* c1, int* ci, const float* rmax, const float* rsum, const float* cmax,
                           const float* csum, const float* l0, const float* l1, const unsigned char* flags, void* ws, size_t ws_bytes, void* stream);

/* ---- dual-softmax matcher (imcui/hloc/matchers/dual_softmax.py; zoo entries disk+dualsoftmax, superpoint+dualsoftmax) */
size_t imcui_hip_dual_softmax_workspace_bytes(int B, int C, int N, int M);
/* desc0 [dev, B,C,N], desc1 [dev, B,C,M] channels-first as the plugin receives them (any C >= 1); the matcher
 * L2-normalises over C when `normalize` (dual_softmax.py:20-22), sim = D0^T D1 * inv_temperature,
 * P = softmax(sim, rows) * softmax(sim, columns); matches0 [dev, B,N] int32 = the last column that is the row
 * maximum, the column maximum and > threshold (-1 = none), scores0 [dev, B,N] = P there (0 otherwise).
 * Per batch item (the reference is only ever driven with B = 1). */
int imcui_hip_dual_softmax(imcui_hip_t* h, const float* desc0, const float* desc1, int B, int C, int N, int M, double threshold,
                           double inv_temperature, int normalize, int* matches0, float* scores0, void* ws, size_t ws_bytes,
                           void* stream);

/* ---- building blocks (exported for tests and for the LoFTR path to come) --------------------- */
/* C[M,N] = A[M,K] * W[N,K]^T + bias ; relu optional ; K % 32 == 0 */
int imcui_hip_linear_f32(imcui_hip_t* h, const float* A, const float* W, const float* bias, float* C, int M, int N, int K,
                         int relu, void* stream);
/* NHWC 3x3 conv (pad 1) + bias (+ReLU) (+2x2 max-pool); weights OIHW on the HOST are packed with
 * imcui_hip_conv3x3_pack (Cin % 32 == 0, Cout % 64 == 0) into Cout*Cin*9 floats. */
int imcui_hip_conv3x3_pack(const float* w_oihw, int Cout, int Cin, float* packed);
int imcui_hip_conv3x3_f32(imcui_hip_t* h, const float* in_nhwc, const float* packed_w, const float* bias, float* out_nhwc,
                          int B, int H, int W, int Cin, int Cout, int relu, int pool, void* stream);
/* split-precision variants of the conv building block (mode 1 packing / launch) */
float imcui_hip_conv3x3_pack_split(const float* w_oihw, int Cout, int Cin, unsigned short* hi, unsigned short* lo);
/* Step before the path: the host preprocessing of imcui/hloc/extract_features.py:120-160 for
 * images that need no resize -- cv2.cvtColor(RGB2GRAY) on 8-bit pixels (OpenCV 4.x fixed point:
 * (9798 R + 19235 G + 3735 B + 16384) >> 15), astype(float32), / 255.0 -- on the device.
 * rgb: uint8 [B,H,W,3] (interleaved, as decoded), out: float32 [B,1,H,W].  H*W % 4 == 0. */
int imcui_hip_rgb_to_gray_f32(imcui_hip_t* h, const unsigned char* rgb_hwc, float* out, int B, int H, int W, void* stream);

/* The same step WITH a resize (imcui/hloc/extract_features.py:26-40 `resize_image(.., "cv2_area")`, :80-99, :120-148):
 *   uint8 [B,H,W,C] (C = 1 gray, C = 3 RGB interleaved) -> [cv2 RGB2GRAY fixed point] -> float32 -> cv2.resize(INTER_AREA)
 *   to oh x ow -> / 255.0 -> out float32 [B,1,oh,ow].  Shrinking only (the reference switches to INTER_LINEAR when a side
 * grows: IMCUI_HIP_ERR_UNSUPPORTED).  Non-integer factors use OpenCV's decimation tables, built on the HOST by
 * imcui_hip_area_table(ssize, dsize, start[dsize+1], index, weight) (returns the entry count; call it with
 * index = weight = NULL first to size the arrays) and uploaded by the caller: x tables for W -> ow, y tables for H -> oh
 * (NULL for integer factors).  float32 arithmetic in OpenCV's order; equals oracle/preprocess.py bit for bit. */
int imcui_hip_area_table(int ssize, int dsize, int* start, int* index, float* weight);
int imcui_hip_preprocess_area_f32(imcui_hip_t* h, const unsigned char* src, int B, int H, int W, int C, const int* xstart,
                                  const int* xindex, const float* xweight, const int* ystart, const int* yindex,
                                  const float* yweight, float* out, int oh, int ow, void* stream);

/* Growing resize of the same preprocessing step: `resize_image(image, size, "cv2_area")` runs cv2.INTER_LINEAR as soon as a
 * side grows (imcui/hloc/extract_features.py:29-31; `superpoint_max` force-resizes every image to 640 x 480).  Host table of
 * OpenCV's float-image set-up (half-pixel centres, float weights): per destination index the two source indices and the weight
 * of the second; horizontal = 1 pins out-of-range taps to the border with weight 0, horizontal = 0 (rows) clips the indices. */
int imcui_hip_linear_table(int ssize, int dsize, int horizontal, int* i0, int* i1, float* w1);
/* uint8 [B][H][W][C] (C = 1 gray, 3 RGB -> cv2's fixed-point gray) -> float32 [B][1][oh][ow] = INTER_LINEAR(float image) / 255:
 * horizontal pass of the two rows, then the vertical weights; multiply, multiply, add in float32 (no fused multiply-add).
 * Tables on the device.  Bit-exact against oracle/preprocess.py: linear_resize_f32 (parity unpinned: cv2 is absent). */
int imcui_hip_preprocess_linear_f32(imcui_hip_t* h, const unsigned char* src, int B, int H, int W, int C, const int* x0, const int* x1,
                                    const float* a1, const int* y0, const int* y1, const float* b1, float* out, int oh, int ow,
                                    void* stream);

/* The dfactor resize of the float image: `F.resize(image, size=size_new, antialias=True)` (extract_features.py:142-148,
 * match_dense.py:182) = ATen's anti-aliased bilinear kernel.  Host table of `_compute_indices_weights_aa` (float32): first
 * source index, tap count and normalised weights [out_size][kmax] per output index; returns the largest tap count (call with
 * w == NULL to size kmax). */
int imcui_hip_aa_table(int in_size, int out_size, int* first, int* count, float* w, int kmax);
/* float32 [planes][H][W] -> [planes][oh][ow]: the width pass of every needed source row (src[0] * w[0], then fused
 * multiply-adds in tap order), then the same along the height -- the arithmetic of ATen's CPU kernel, bit for bit
 * (tests/test_oracle_preprocess.py pins the restatement to torch; tests/test_gpu_preprocess.py the kernel). */
int imcui_hip_resize_aa_f32(imcui_hip_t* h, const float* src, int planes, int H, int W, const int* xfirst, const int* xcount,
                            const float* xweight, int kx, const int* yfirst, const int* ycount, const float* yweight, int ky, float* out,
                            int oh, int ow, void* stream);

/* nn.Linear weight [N][K] (K % 16 == 0) -> f16 hi / lo planes of w * 2^e in the FRAGMENT-MAJOR order the split GEMM
 * streams ([ceil(N/32)][K/16][2][32][8] halves per plane, rows >= N zero: each 1 KiB block is one MFMA operand
 * fragment of a wave); returns 2^-e (0 on bad arguments).  Planes hold roundup(N,32) * K halves each. */
float imcui_hip_linear_pack_split(const float* w, int N, int K, unsigned short* hi, unsigned short* lo);
/* C = act(A[M,K] * W^T + bias) with pre-split weights (device planes from imcui_hip_linear_pack_split, wscale = device
 * float holding the returned 2^-e).  Building block of every network projection in the split mode
 * (imcui/hloc/matchers/lightglue.py:75 -> upstream nn.Linear); precision 1 only, K % 32 == 0. */
int imcui_hip_linear_split_f32(imcui_hip_t* h, const float* A, const unsigned short* wh, const unsigned short* wl,
                               const float* wscale, const float* bias, float* C, int M, int N, int K, int relu, void* stream);

/* Opt-in range check of the split arithmetic (debugging aid; also switched on by the environment variable
 * IMCUI_HIP_CHECK_RANGE=1 at imcui_hip_create).  The 3 x f16 split of an f32 activation saturates its hi part at 65504 and is
 * no longer fp32-grade above that; weights are rescaled at pack time, activations are not.  With the check enabled every
 * split-mode GEMM / 3x3 convolution / fused-FFN launch first scans its f32 activation operand.  imcui_hip_get_range_status
 * synchronises the device, returns the accumulated word (bit 0: a value beyond the f16 range, bit 1: NaN / Inf) and clears it. */
int imcui_hip_set_range_check(imcui_hip_t* h, int enable);
int imcui_hip_get_range_status(imcui_hip_t* h, int* status);

/* Projection of a transformer block into the attention kernels' operand layout (upstream LightGlue SelfBlock `Wqkv` + rotary
 * encoding, CrossBlock `to_qk` / `to_v`; reached from imcui/hloc/matchers/lightglue.py:75): x [nseq * rows_per_seq][256] ->
 * f16 hi / lo planes (hi plane first, lo plane nseq*rows_per_seq*256 halves behind it) of Q, K [seq][head][row][64] and of
 * V^T [seq][head][64][row], 4 heads.  cross == 0: W [768][256] packed rows (q | k | v), rotary encoding (cos / sin
 * [rows][32]) on q and k, q *= alpha.  cross == 1: W [512][256] rows (qk | v), no rotation, qk *= alpha, `k` unused.
 * Weights as planes from imcui_hip_linear_pack_split; precision 1 only; rows_per_seq % 128 == 0; row tiles whose first row is
 * >= cnt[seq] are skipped.  Building block of the parity tests (both GEMM kernels that implement it are compared). */
int imcui_hip_qkv_split_f32(imcui_hip_t* h, const float* x, const unsigned short* wh, const unsigned short* wl, const float* wscale,
                            const float* bias, const float* rope_cos, const float* rope_sin, const int* cnt, int nseq,
                            int rows_per_seq, float alpha, int cross, float* q, float* k, float* v, void* stream);

/* Fused transformer FFN of a LightGlue block (upstream TransformerLayer.ffn on cat([x, message]) with the attention
 * out-projection folded into W1; reached from imcui/hloc/matchers/lightglue.py:75):
 *   out = x + W2 * GELU(LayerNorm_512(W1 * [x | ctx] + b1)) + b2,   x, ctx, out [M][256] (out may alias x).
 * W1 [512][512] as planes from imcui_hip_linear_pack_split; W2 [256][512] as planes from imcui_hip_ffn_pack_w2 (the
 * same fragment-major planes with the K axis in the order the kernel's first GEMM hands its accumulators over);
 * s1 / s2 = device floats holding the returned 2^-e.  precision 1 only.  One kernel: the 512-wide hidden row stays
 * on the CU (registers -> LDS), no LayerNorm / GELU pass over HBM.  act = 1 replaces LayerNorm + GELU by ReLU (gamma, beta
 * may be NULL): SuperGlue's MLP([512, 512, 256]) with its BatchNorm folded into W1 / b1 (imcui/hloc/matchers/superglue.py:26).
 * act = 2 / 3: out = x + LayerNorm_256(W2 * act(W1 * [x | ctx] + b1) + b2) with LeakyReLU(0.01) / ReLU and gamma, beta [256]
 * (b1, b2 may be NULL): the coarse MLPs of EfficientLoFTR (eloftr.py:79) / LoFTR (loftr.py:54). */
float imcui_hip_ffn_pack_w2(const float* w2, unsigned short* hi, unsigned short* lo);
int imcui_hip_ffn_split_f32(imcui_hip_t* h, const float* x, const float* ctx, const unsigned short* w1h, const unsigned short* w1l,
                            const float* s1, const float* b1, const float* gamma, const float* beta, const unsigned short* w2h,
                            const unsigned short* w2l, const float* s2, const float* b2, float* out, int M, int act, void* stream);

/* Lab hook (tools/ffn_bench.py): when `stamps` is non-NULL every workgroup of the next imcui_hip_ffn_split_f32 calls
 * writes 8 wall-clock (100 MHz) values at its phase boundaries to stamps[8 * workgroup ..]; NULL switches it off. */
int imcui_hip_ffn_set_debug(imcui_hip_t* h, long long* stamps);

int imcui_hip_conv3x3_split_f32(imcui_hip_t* h, const float* in_nhwc, const unsigned short* wh, const unsigned short* wl,
                                const float* wscale, const float* bias, float* out_nhwc, int B, int H, int W, int Cin,
                                int Cout, int relu, int pool, void* stream);
/* NHWC convolution as an implicit-im2col GEMM: weights [Cout][k*k*Cin] (tap-major), Cin % 32 == 0,
 * optional residual [B,Hout,Wout,Cout] added before the activation (0 none, 1 ReLU, 2 LeakyReLU 0.01) */
int imcui_hip_conv_gemm_f32(imcui_hip_t* h, const float* in_nhwc, const float* w_gemm, const float* bias, const float* resid,
                            float* out_nhwc, int B, int Hin, int Win, int Cin, int Cout, int ksize, int stride, int act,
                            void* stream);
/* softmax(Q K^T) V, head_dim 64, operands head-major [S][heads][rows][64] (Q pre-scaled),
 * output token-major [S*rows][heads*64]; cnt [dev, S] valid rows; cross: keys of sequence s^1.
 * log2_domain = 1: Q additionally carries a factor log2(e) and the soft-max is evaluated in base 2 (what the LightGlue /
 * SuperGlue layers do: one v_exp_f32 per probability, no multiply); 0: natural-log operands. */
int imcui_hip_attention_f32(imcui_hip_t* h, const float* Q, const float* K, const float* V, float* O, const int* cnt, int S,
                            int heads, int rows, int cross, int log2_domain, void* stream);
/* The same block in attention variant 9 (round 6; split mode, log2 domain only: what LightGlue's layers run when option "attn_variant" = 9):
 * K.Q^T in three f16 products, P.V as one f16 product + two block-scaled fp6 correction products (csrc/attention_mx.hip).  Operands as
 * for imcui_hip_attention_f32 in mode 1 (pre-split planes; the lo plane of V^T is read once, to build the fp6 planes); `scratch` [dev,
 * imcui_hip_attention_mx_scratch_bytes(S, heads, rows)] receives the fp6 planes of V^T. */
size_t imcui_hip_attention_mx_scratch_bytes(int S, int heads, int rows);
int imcui_hip_attention_mx_f32(imcui_hip_t* h, const float* Q, const float* K, const float* V, float* O, const int* cnt, int S,
                               int heads, int rows, int cross, void* scratch, size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IMCUI_HIP_H */
