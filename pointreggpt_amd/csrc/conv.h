// conv.h — launch interface of the MFMA convolutions and their weight packing.
#pragma once
#include <vector>

#include "common.h"

namespace prg {

// K-chunk of the generic kernel (elements of one tap's channel range staged per main-loop iteration): 64 bytes
// per tile row.  The halo kernel consumes two of these per step (128-byte pixel rows).
template <typename T>
struct ConvTile {
  static constexpr int BK = 64 / (int)sizeof(T);  // bf16: 32, f32: 16
};

struct ConvDesc {
  int B, Hin, Win;          // source tensors' spatial size (before the optional x2 nearest upsample)
  int C0, C1;               // channels of src0 / src1 (virtual concat [src0, src1]); C1 = 0 when unused
  int ups;                  // 1: conv runs on the 2x nearest-upsampled image (Upsample, sd:592-594)
  int KH, KW, stride, pad;
  int Hout, Wout;
  int Cout, CoutPad;        // CoutPad = Cout rounded up to 64: packed weight rows (zeros beyond Cout)
  int kchunks;              // ceil((C0+C1) / BK)
};

// Packed weights: [tap = kh*KW+kw][chunk][CoutPad][BK] of T, zero padded in both Cout and channel.
template <typename T>
void pack_conv_weight(const float* w_oihw, int Cout, int Cin, int KH, int KW, std::vector<T>& out, int* CoutPad,
                      int* kchunks);

template <typename T>
struct ConvLaunch {
  ConvDesc d;
  const T* src0;
  const T* src1;
  const T* w;            // packed
  const float* bias;     // [Cout] or null
  const T* residual;     // NHWC (M, Cout) added in the epilogue, or null
  // Activated residual (1x1 convs on the implicit-GEMM kernel, bf16 mode): out = conv + bias + SiLU(residual * res_a[b][c]
  // + res_b[b][c]) — the ResnetBlock tail `SiLU(GroupNorm(h)) + res_conv(x)` (sd:731-734) in the res_conv's epilogue, where
  // `residual` is h and (res_a, res_b) are its folded GroupNorm coefficients.  out may alias residual.  null = plain add.
  const float* res_a;
  const float* res_b;
  T* out;                // NHWC (M, Cout)
  // fused GroupNorm statistics of the OUTPUT (before any activation): per (image, M-tile, group) (sum, sumsq)
  // written to gn_partials [B][gn_nsplit][gn_groups][2]; null = off.  launch_conv decides whether the shape
  // allows it and reports the slab count through *gn_nsplit_out (0 = caller must run the stats kernel).
  float* gn_partials;
  int gn_groups;
  // fused prologue on the INPUT (halo kernel, single source): x <- silu(x * pro_a[b][c] + pro_b[b][c]);
  // this is GroupNorm + (scale+1, shift) + SiLU of the previous Block folded into per-(image, channel) affine
  // coefficients (sd:690-696).  null = off.
  const float* pro_a;
  const float* pro_b;
  // Optional in-kernel coefficient folding (kernels that support it report so through launch_conv's coef_done): the
  // workgroup that completes an image's last tile turns the image's partial sums into gn_coef_a/b [B][Cout]
  // (= what gn_coeff_kernel would compute in a launch of its own).  gn_tickets: [B] ints, zero between launches.
  GnApply gn;
  float* gn_coef_a;
  float* gn_coef_b;
  int* gn_tickets;
  // Fixed-point statistics (common.h, GnFold): producers that support it add their wave totals into gn_acc [B][gn_groups][2]
  // (zeroed by the caller) INSTEAD of writing gn_partials slabs and report so through launch_conv's acc_done; consumers
  // with pro_fold.acc != null compute the prologue coefficients themselves (pro_a / pro_b then point to [B][C0] scratch
  // tables that kernels without the in-kernel fold fill with one gn_coeff_acc launch first).
  long long* gn_acc;
  GnFold pro_fold;
  // res_fold.acc != null (with `residual`, 1x1 convs on the implicit-GEMM kernel): the activated residual's coefficients are
  // folded in the epilogue from the accumulators (once per thread and image) instead of read from res_a / res_b
  GnFold res_fold;
  // MX-fp8 operands (handles of dtype PRG_MXFP8, 3x3 / s1 / p1 convs with 64-channel multiples): OCP e4m3 weights
  // [tap][64-channel chunk][CoutPad][64] with one E8M0 scale per 32 input channels [tap][chunk][CoutPad][2]; null = bf16.
  const uint8_t* w_mx;
  const uint8_t* w_mx_scale;
  int mx_pure;           // 1: never fall back to bf16 operands for shapes the 256-pixel MX kernel does not cover (unit tests)
  // Conv2d(C, Cout, 4, stride 2, pad 1) only: the same weights as the 3 x 3 packing of the equivalent 2 x 2-tap
  // convolution over the space-to-depth view of the shifted input ([Cout][4 C][3][3], taps (1..2, 1..2) non-zero; virtual
  // channel (2 dy + dx) C + c, tap (1 + by, 1 + bx) = W[.][c][2 by + dy][2 bx + dx]); null = not available.  conv_w256.hip
  const T* w_s2d;
  int s2d_kchunks;
  // f16x3 mode (handles of dtype PRG_F16X3, T = float; conv_split.hip): the weights split into f16 hi / lo halves,
  // [tap][32-channel chunk][CoutPad][32 hi | 32 lo]; null = the exact-f32 kernels
  const uint16_t* w_split;
  int split_kchunks;
};

// hi / lo f16 split of a conv weight for the f16x3 mode (layout at ConvLaunch::w_split)
void pack_conv_weight_split(const float* w_oihw, int Cout, int Cin, int KH, int KW, std::vector<uint16_t>& out, int* CoutPad,
                            int* kchunks32);
// f16x3 kernels: 1 = launched, 0 = shape not covered (the exact-f32 kernels run), < 0 = error
int try_launch_conv_split(const ConvLaunch<float>& L, hipStream_t s, int* gn_nsplit_out);

// OIHW weights of the equivalent convolution described at ConvLaunch::w_s2d, from Conv2d(Cin, Cout, 4, 2, 1) weights
void s2d_equivalent_weights(const float* w_oihw, int Cout, int Cin, std::vector<float>& out);

// MX (OCP microscaling) fp8 packing of a conv weight: per (tap, output channel) the input channels are cut into blocks of
// 32; block scale = 2^(floor(log2 max|w|) - 8) as an E8M0 byte (bias 127), elements = e4m3fn(w / scale), round to
// nearest even, saturating at 448.  data: [tap][ceil(Cin/64)][CoutPad][64], scales: [tap][ceil(Cin/64)][CoutPad][2].
void pack_conv_weight_mxfp8(const float* w_oihw, int Cout, int Cin, int KH, int KW, std::vector<uint8_t>& data,
                            std::vector<uint8_t>& scales, int* CoutPad, int* chunks64);
// host-side e4m3fn conversion used by the packer (and, through the debug entry, checked against the device's)
uint8_t f32_to_e4m3(float v);
float e4m3_to_f32(uint8_t b);

// Returns PRG_OK; *gn_nsplit_out (may be null) receives the number of statistic slabs per image written, or 0
// when statistics were requested but this shape cannot fuse them.  `allow_prologue` must be checked by the
// caller with conv_supports_prologue() before setting pro_a / pro_b.
template <typename T>
int launch_conv(const ConvLaunch<T>& L, hipStream_t s, int* gn_nsplit_out, int* coef_done = nullptr, int* acc_done = nullptr);

// true when the 3x3 halo kernel will run this conv (so a fused input prologue is available)
template <typename T>
bool conv_supports_prologue(const ConvDesc& d);

inline double conv_flops(const ConvDesc& d) {
  return 2.0 * (double)d.B * d.Hout * d.Wout * d.Cout * (double)(d.C0 + d.C1) * d.KH * d.KW;
}

// Direct (VALU) stem conv 7x7 pad 3: float32 NCHW (B,Cin,H,W) with Cin in {1,3} -> T NHWC (B,H,W,Cout).
// wk = float32 [49*Cin][Cout] (tap-major, then cin), bias float32 [Cout].
// stem on MFMA (bf16 path, Cin 1 -> 64): fragments packed once by pack_stem_mfma_weights
bool stem_conv_mfma_supported(int Cin, int Cout, int H, int W);
int launch_stem_conv_mfma(const float* x, const bf16_t* wf, const float* bias, bf16_t* out, int B, int H, int W, hipStream_t s);
void pack_stem_mfma_weights(const float* w, std::vector<bf16_t>& out);
template <typename T>
int launch_stem_conv(const float* x, const float* wk, const float* bias, T* out, int B, int Cin, int H, int W,
                     int Cout, hipStream_t s);

// Head: 1x1 conv to ONE channel (+ optional sigmoid): T NHWC (M, C) -> float32 (M,).
template <typename T>
int launch_head_conv(const T* x, const float* w, const float* bias, float* out, int64_t M, int C, int sigmoid,
                     hipStream_t s);

}  // namespace prg
