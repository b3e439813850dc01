// NHWC f32 convolutions for the SuperPoint VGG encoder / heads (gfx950).
#pragma once
#include "common.h"

// 3x3, pad 1, stride 1, +bias, +ReLU, optional fused 2x2/2 max-pool.
// in  : [B, H, W, Cin]  (NHWC), Cin % 32 == 0
// wp  : packed weights  [Cin/32][9 taps][8 cq][Cout][4]   (see pack_conv3x3)
// out : [B, H, W, Cout] or, with pool, [B, H/2, W/2, Cout];  Cout % 64 == 0
int conv3x3_launch(imcui_hip_s* h, const float* in, const float* wp, const float* bias, float* out, int B, int H, int W,
                   int Cin, int Cout, int relu, int pool, hipStream_t stream);

// DPT head fused into the epilogue of its last 3x3 convolution (128 channels, ReLU): 1x1 convolution to 4 channels (w [4][128], b [4])
// + point-map post-processing -> pts [B*H*W][3] = xyz / |xyz| * expm1(|xyz|), conf [B*H*W] = 1 + exp(c), raw [B*H*W][4] optional.
// With a head the 128-channel map is written only if `out` is not null.
struct ConvHead {
    const float* w = nullptr;
    const float* b = nullptr;
    float* pts = nullptr;
    float* conf = nullptr;
    float* raw = nullptr;
};

// split-precision variant (imcui_hip_s::precision == 1): weights pre-split into f16 hi / lo planes
//   wh/wl : [Cin/32][9 taps][4 octets][Cout][8 halves] of w * 2^e ; wscale -> 2^-e (device scalar)
//   relu  : activation code 0 none / 1 ReLU / 2 LeakyReLU(0.01);  resid (optional, no pooling): a map of the output's shape
//           added before the activation (residual blocks of the dense matchers' backbones)
int conv3x3_split_launch(imcui_hip_s* h, const float* in, const unsigned short* wh, const unsigned short* wl,
                         const float* wscale, const float* bias, float* out, int B, int H, int W, int Cin, int Cout,
                         int relu, int pool, hipStream_t stream, const float* resid = nullptr, int cin_stride = 0, int cout_live = 0,
                         int single = 0, const float* resid2 = nullptr, const ConvHead* head = nullptr);
// relu bit 2 (value 4): ReLU applied to the INPUT map while it is staged (the producer left it un-activated);  resid2: a second map
// added after resid (out = act(conv + resid + resid2))
// cin_stride: floats between two pixels of `in` when the map stores more channels than the Cin that are used (0 = Cin)
// cout_live: output channels >= cout_live are zero padding of the layer (zero weights and bias): their 32-channel fragments
// are not multiplied, the channels are stored as zeros (0 = Cout)
// single: one f16 product per element pair (hi planes only) instead of three, see GemmP.single
// SuperPoint conv1a (1->64, VALU, evaluated on the fly for the patch) fused into conv1b (64->64, split MFMA):
// image [B,H,W] -> relu(conv1b(relu(conv1a(image)))) (+2x2 max-pool), NHWC out
int conv1ab_fused_split_launch(imcui_hip_s* h, const float* image, const float* w1a, const float* b1a,
                               const unsigned short* wh, const unsigned short* wl, const float* wscale, const float* bias,
                               float* out, int B, int H, int W, int pool, hipStream_t stream);
// host: OIHW -> the split layout above; returns 2^-e
float pack_conv3x3_split(const float* w_oihw, int Cout, int Cin, unsigned short* hi, unsigned short* lo);
// host: the same from the implicit-GEMM layout [Cout][9 taps][Cin] (pack_conv_gemm); planes of 9 * Cin * Cout halves each
float pack_conv3x3_split_from_gemm(const float* w_gemm, int Cout, int Cin, unsigned short* hi, unsigned short* lo, int cin_used = 0);

// first layer: 1 -> 64 channels, 3x3, pad 1, +bias, +ReLU.  in [B,H,W] ; w [9][64] ; out [B,H,W,64]
int conv1a_launch(imcui_hip_s* h, const float* in, const float* w, const float* bias, float* out, int B, int H, int W,
                  hipStream_t stream);

// host-side packers (OIHW -> device layouts above)
void pack_conv3x3(const float* w_oihw, int Cout, int Cin, float* dst);
void pack_conv1a(const float* w_oihw, float* dst);  // [64,1,3,3] -> [9][64]
