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
  // Upsample convs (nn.Upsample(x2, nearest) + Conv2d(C, Cout, 3, pad 1), d.ups = 1; bf16): the four pre-summed 2 x 2-tap packings of
  // the sub-pixel decomposition (conv_w256.hip, MODE 2), [phase 2 dy + dx][tap 2 a + b][32-channel chunk][CoutPad][32]; null = the
  // nine-tap gather form.  up_equivalent_weights() builds the OIHW tensors.
  const T* w_up = nullptr;
  // f16x3 mode (handles of dtype PRG_F16X3, T = float; conv_split.hip): the weights split into f16 hi / lo halves,
  // [tap][32-channel chunk][CoutPad][32 hi | 32 lo]; null = the exact-f32 kernels
  const uint16_t* w_split;
  int split_kchunks;
  // f16x3, Upsample convs: the four sub-pixel 2 x 2-tap packings (up_equivalent_weights) in the split layout, phase-major:
  // [phase][tap 4][32-channel chunk][CoutPad][32 hi | 32 lo]; null = the nine-tap gather form
  const uint16_t* w_up_split = nullptr;
  // f16x3: [CoutPad] (w_split) / [4][CoutPad] (w_up_split, phase-major) exact power-of-two factors applied to the float32 totals
  // before the bias (the packer scaled every output channel's weights by the inverse); null = 1
  const float* split_scale = nullptr;
  const float* split_scale_up = nullptr;
  // h16 (bf16 mode, round 4): the tensor between the two convs of a ResnetBlock only ever feeds conv2's fused GroupNorm + SiLU
  // prologue, so it is stored as IEEE f16 instead of bf16 — the prologue then runs in PACKED f16 (v_pk_fma_f16, v_exp_f16,
  // v_rcp_f16: two channels per instruction, no unpack / repack) and conv2 contracts f16 operands
  // (v_mfma_f32_32x32x16_f16, the bf16 instruction's rate) against an f16 packing of its weights.
  //   out_f16: this launch stores f16 bit patterns into `out` (conv1);  in_f16: src0 holds f16, the prologue is pro_fold's and
  //   w_f16 is the f16 twin of `w`, same layout (conv2).  Only the kernels that conv_h16_pair_ok() probes implement them.
  //   probe: try_launch_* return 1 where they would launch, without launching.
  int out_f16 = 0, in_f16 = 0, probe = 0;
  const uint16_t* w_f16 = nullptr;
};

#if defined(__HIPCC__)
typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
// two floats -> packed f16 (v_cvt_pk_f16_f32, round to nearest even): the h16 output format
__device__ inline uint32_t h16_pack(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  const f32x2 p = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(p, h16x2));
}
// GroupNorm affine + SiLU on one 16-byte unit = four packed f16 channel pairs (the h16 input format).  Per pair: v_pk_fma_f16,
// v_pk_mul_f16, two v_exp_f16 (the second writes the upper half of the same register through SDWA: no repack), v_pk_add_f16, two
// v_rcp_f16, v_pk_mul_f16 — eight instructions against fourteen on the bf16 / float32 path.  exp2 overflow (y < -11) ends in
// rcp(inf) = 0 = SiLU's limit.  The transcendentals are inline asm (hipcc repacks with v_pack_b32_f16 otherwise), which hides
// them from the hazard recogniser: gfx950 needs one wait state between a transcendental / a dst_sel write and a VALU consumer
// of its result (LLVM: hasTransForwardingHazard, hasDstSelForwardingHazard) — the trailing s_nop of each block.
typedef __attribute__((ext_vector_type(4))) unsigned int h16_u32x4;
#define PRG_H16_TRANS4(OP)                                                                                             \
  asm(OP "_e32 %0, %4\n\t" OP "_e32 %1, %5\n\t" OP "_e32 %2, %6\n\t" OP "_e32 %3, %7\n\t"                            \
      OP "_sdwa %0, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"                                  \
      OP "_sdwa %1, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"                                  \
      OP "_sdwa %2, %6 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"                                  \
      OP "_sdwa %3, %7 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\ts_nop 0"                           \
      : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)                                                                      \
      : "v"(i0), "v"(i1), "v"(i2), "v"(i3))
__device__ inline h16_u32x4 h16_silu8(h16_u32x4 x, const h16x2 (&a)[4], const h16x2 (&b)[4]) {
  const h16x2 nl2e = {(_Float16)-1.4426950408889634f, (_Float16)-1.4426950408889634f}, one = {(_Float16)1.0f, (_Float16)1.0f};
  h16x2 y[4];
  uint32_t i0, i1, i2, i3, o0, o1, o2, o3;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t w = x[j];   // (hipcc 7.2: __builtin_bit_cast of the vector ELEMENT x[j] reads element 0 for every j)
    y[j] = __builtin_elementwise_fma(__builtin_bit_cast(h16x2, w), a[j], b[j]);
  }
  i0 = __builtin_bit_cast(uint32_t, y[0] * nl2e); i1 = __builtin_bit_cast(uint32_t, y[1] * nl2e);
  i2 = __builtin_bit_cast(uint32_t, y[2] * nl2e); i3 = __builtin_bit_cast(uint32_t, y[3] * nl2e);
  PRG_H16_TRANS4("v_exp_f16");
  i0 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h16x2, o0) + one); i1 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h16x2, o1) + one);
  i2 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h16x2, o2) + one); i3 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h16x2, o3) + one);
  PRG_H16_TRANS4("v_rcp_f16");
  h16_u32x4 r;
  r[0] = __builtin_bit_cast(uint32_t, y[0] * __builtin_bit_cast(h16x2, o0));
  r[1] = __builtin_bit_cast(uint32_t, y[1] * __builtin_bit_cast(h16x2, o1));
  r[2] = __builtin_bit_cast(uint32_t, y[2] * __builtin_bit_cast(h16x2, o2));
  r[3] = __builtin_bit_cast(uint32_t, y[3] * __builtin_bit_cast(h16x2, o3));
  return r;
}
#undef PRG_H16_TRANS4
#endif

// f16 twin of pack_conv_weight<bf16_t> (same [tap][32-channel chunk][CoutPad][32] layout, IEEE f16 bits, round to nearest even)
void pack_conv_weight_f16(const float* w_oihw, int Cout, int Cin, int KH, int KW, std::vector<uint16_t>& out);
// true when conv1 (L1, asked to store f16) and conv2 (L2, asked to read f16 through its folded prologue) of a ResnetBlock would both
// run on kernels that implement the h16 format, L1 with fixed-point statistics; nothing is launched
bool conv_h16_pair_ok(const ConvLaunch<bf16_t>& L1, const ConvLaunch<bf16_t>& L2);

// hi / lo f16 split of a conv weight for the f16x3 mode (layout at ConvLaunch::w_split)
// oscale (may be null = no scaling): receives [CoutPad] exact powers of two the kernels multiply the float32 totals by — the
// inverse of the per-output-channel scale the packer applied so that the lo halves stay normal f16 (ConvLaunch::split_scale)
void pack_conv_weight_split(const float* w_oihw, int Cout, int Cin, int KH, int KW, std::vector<uint16_t>& out, int* CoutPad,
                            int* kchunks32, std::vector<float>* oscale = nullptr);
// f16x3 kernels: 1 = launched, 0 = shape not covered (the exact-f32 kernels run), < 0 = error
int try_launch_conv_split(const ConvLaunch<float>& L, hipStream_t s, int* gn_nsplit_out);

// OIHW weights of the equivalent convolution described at ConvLaunch::w_s2d, from Conv2d(Cin, Cout, 4, 2, 1) weights
void s2d_equivalent_weights(const float* w_oihw, int Cout, int Cin, std::vector<float>& out);
// The four [Cout][Cin][2][2] tensors (phase-major) of ConvLaunch::w_up from Conv2d(Cin, Cout, 3, 1, 1) weights: for output sub-pixel
// dy the kernel rows {0 | 1, 2} (dy = 0) or {0, 1 | 2} (dy = 1) fall on the two source rows of its window (summed in float64)
void up_equivalent_weights(const float* w_oihw, int Cout, int Cin, std::vector<float>& out);

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

// executed / algorithmic MAC ratio of the calling thread's last launch_conv: 1, or 4 / 9 when an Upsample conv ran as four
// 2 x 2-tap sub-pixel convolutions (conv_w256.hip MODE 2) — conv_flops() below is the ALGORITHMIC count (the reference's op)
double conv_last_exec_scale();
int conv_last_was_mx();

inline double conv_flops(const ConvDesc& d) {
  return 2.0 * (double)d.B * d.Hout * d.Wout * d.Cout * (double)(d.C0 + d.C1) * d.KH * d.KW;
}

// Direct (VALU) stem conv 7x7 pad 3: float32 NCHW (B,Cin,H,W) with Cin in {1,3} -> T NHWC (B,H,W,Cout).
// wk = float32 [49*Cin][Cout] (tap-major, then cin), bias float32 [Cout].
// stem on MFMA (bf16 path, Cin 1 or 3 -> 64): fragments packed once by pack_stem_mfma_weights
bool stem_conv_mfma_supported(int Cin, int Cout, int H, int W);
int launch_stem_conv_mfma(const float* x, const bf16_t* wf, const float* bias, bf16_t* out, int B, int Cin, int H, int W, hipStream_t s);
void pack_stem_mfma_weights(const float* w, int Cin, std::vector<bf16_t>& out);
// f16x3 (round 5): the same kernel on f16 hi / lo halves with a float32 output; scale[64] = the inverse of the packer's per-channel power of two
int launch_stem_conv_mfma_split(const float* x, const uint16_t* wf, const float* bias, const float* wscale, float* out, int B, int Cin,
                                int H, int W, hipStream_t s);
void pack_stem_mfma_weights_split(const float* w, int Cin, std::vector<uint16_t>& out, std::vector<float>& scale);
template <typename T>
int launch_stem_conv(const float* x, const float* wk, const float* bias, T* out, int B, int Cin, int H, int W,
                     int Cout, hipStream_t s);

// Head: 1x1 conv to ONE channel (+ optional sigmoid): T NHWC (M, C) -> float32 (M,).
template <typename T>
int launch_head_conv(const T* x, const float* w, const float* bias, float* out, int64_t M, int C, int sigmoid,
                     hipStream_t s);

}  // namespace prg
