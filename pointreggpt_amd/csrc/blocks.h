// blocks.h — launch interface of the non-conv U-Net kernels (all HBM-bound): GroupNorm statistics / apply,
// channel LayerNorm, linear attention, full attention, the small conditioning MLPs.
#pragma once
#include "common.h"

namespace prg {

constexpr int kHeads = 4, kDimHead = 32, kHidden = 128;   // sd:738, sd:773

// GroupNorm (sd:685, eps 1e-5) over NHWC x (B, HW, C).  Pass 1 writes per-slab (sum, sumsq) partials
// [B][nsplit][G][2]; pass 2 reduces them in a fixed order (deterministic), normalises, applies the optional
// conditioning (scale+1, shift) (sd:692-694), SiLU, and an optional residual add (sd:734).
// Returns nsplit chosen by pass 1 through *nsplit.
template <typename T>
int launch_gn_stats(const T* x, float* partials, int B, int HW, int C, int G, int* nsplit, hipStream_t s);

// (struct GnApply lives in common.h: the conv kernels that fold the coefficients themselves need it too)
// Fold GroupNorm (+ conditioning) into per-(image, channel) affine coefficients y = x * A + B (the SiLU that follows
// is applied by the consumer): the fused prologue of the next conv reads these.  A, Bc: [B][C] float32.
int launch_gn_coeff(const float* partials, int nsplit, const GnApply& p, float* A, float* Bc, int B, int HW, int C,
                    int G, hipStream_t s);

// The same coefficients from the fixed-point accumulators of the bf16 path (common.h, GnFold) into [B][C] tables: only
// for consumers that cannot fold in-kernel (the wave-specialised conv's prologue, the 1x1 res_conv epilogue).
int launch_gn_coeff_acc(const GnFold& f, float* A, float* Bc, int B, int C, hipStream_t s);
// P = gamma * (scale + 1), Q = beta * (scale + 1) + shift for every conditioned GroupNorm of a forward, ONE launch:
// entries [n] of (ss_off, C, gamma offset, beta offset into `flat`); pq [B][ss_total]: P at ss_off, Q at ss_off + C.
struct CondFoldEntry { int ss_off, C; long long gamma_off, beta_off; };
int launch_cond_fold(const CondFoldEntry* entries, int n, const float* flat, const GnApply& ss, float* pq, int64_t pq_stride,
                     int B, hipStream_t s, long long* zero = nullptr, int64_t zero_words = 0);
// y = silu(x * A + B) + residual with the coefficients folded per block from the accumulators (one image per block row)
int launch_affine_silu_fold(const bf16_t* x, const GnFold& f, const bf16_t* residual, bf16_t* out, int B, int HW, int C,
                            hipStream_t s);

// y = silu(x * A[b][c] + Bc[b][c]) + residual over NHWC x (B, HW, C): the apply half of GroupNorm once gn_coeff ran.
template <typename T>
int launch_affine_silu(const T* x, const float* A, const float* Bc, const T* residual, T* out, int B, int HW, int C,
                       hipStream_t s);

template <typename T>
int launch_gn_apply(const T* x, const float* partials, int nsplit, const GnApply& p, const T* residual, T* out, int B,
                    int HW, int C, int G, hipStream_t s);

// Channel LayerNorm with gain (sd:619-628) + optional residual add (sd:589).
template <typename T>
int launch_layernorm(const T* x, const float* g, const T* residual, T* out, int64_t M, int C, hipStream_t s);

// LinearAttention core (sd:755-768) on qkv NHWC (B, N, 384) -> out NHWC (B, N, 128).
// ws: float workspace of at least linattn_ws_floats(B, N) floats.
size_t linattn_ws_floats(int B, int N);
template <typename T>
int launch_linear_attention(const T* qkv, T* out, float* ws, int B, int N, hipStream_t s);

// Residual(PreNorm(LinearAttention)) fused for the bf16 path (attn_fused.hip): x, out (B, N, C); wqkv [384][C] bf16 with
// the PreNorm gain folded in, wout [C][128] bf16.  ws: at least linattn_fused_ws_floats(B, N) floats.
// f16x3 mode (attn_split.hip): fused Residual(PreNorm(LinearAttention)) on float32 activations, split-f16 contractions
bool linattn_split_supported(int C, int N);
size_t linattn_split_ws_floats(int B, int N);
int launch_linear_attention_split(const float* x, const uint16_t* wqkv_h, const uint16_t* wqkv_l, const uint16_t* wout_h,
                                  const uint16_t* wout_l, const float* bias, const float* out_g, float* out, float* ws, int B, int N,
                                  int C, hipStream_t s);
// f16x3 mode: the bottleneck attention core (sd:789-795) on split-f16 MFMAs, float32 in / out (attn_split.hip)
bool full_attention_split_supported(int N);
int launch_full_attention_split(const float* qkv, float* out, int B, int N, hipStream_t s);
bool linattn_fused_supported(int C);
size_t linattn_fused_ws_floats(int B, int N);
// kshift: [128 + 4] static softmax shifts — a bound on |k| per column, then a bound on |q| per head (see unet.hip) — or
// null = measure the maxima.
int launch_linear_attention_fused(const bf16_t* x, const bf16_t* wqkv, const bf16_t* wout, const float* bias,
                                  const float* out_g, bf16_t* out, float* ws, int B, int N, int C, const float* kshift,
                                  hipStream_t s);

// ResnetBlock tail with the 1x1 res_conv folded in (attn_fused.hip), bf16 path.  wres: [Cout][C0+C1] bf16.
bool resblock_tail_fused_supported(int C0, int C1, int Cout);
// head_out != null (Cout = 64 only): the Unet's final 1x1 conv to one channel (+ optional sigmoid) is applied to the tile
// in LDS and ONLY its float32 result (B, N) is written — the 64-channel tensor never reaches HBM.
// fold != null (acc set): the GroupNorm coefficients are computed per block from the accumulators instead of read from A / Bc.
int launch_resblock_tail_fused(const bf16_t* h, const float* A, const float* Bc, const bf16_t* s0, int C0, const bf16_t* s1,
                               int C1, const bf16_t* wres, const float* bres, bf16_t* out, int B, int N, int Cout,
                               hipStream_t s,
                               const float* head_w = nullptr, const float* head_b = nullptr, float* head_out = nullptr,
                               int head_sigmoid = 0, const GnFold* fold = nullptr);

// Attention core on MFMA (bf16 path, N in {64, 128, 256} tokens) (attn_fused.hip)
bool full_attention_mfma_supported(int N);
int launch_full_attention_mfma(const bf16_t* qkv, bf16_t* out, int B, int N, hipStream_t s);

// Attention core (sd:789-795) on qkv NHWC (B, N, 384) -> out NHWC (B, N, 128).
template <typename T>
int launch_full_attention(const T* qkv, T* out, int B, int N, hipStream_t s);

// y[r][o] = act_out( sum_i act_in(x[r][xoff + i]) * W[o][woff + i] + bias[o] ),  float32, tiny.
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2 };
int launch_linear(const float* x, int ldx, int xoff, const float* W, int ldw, int woff, const float* bias, float* y,
                  int ldy, int R, int I, int O, int act_in, int act_out, hipStream_t s);
// sinusoidal embedding (sd:645-657): t (R,) int64 -> (R, dim) float32
int launch_sinusoidal(const int64_t* t, const float* freqs, float* out, int R, int dim, hipStream_t s);
int launch_sinusoidal_i32(const int32_t* t, const float* freqs, float* out, int R, int dim, hipStream_t s);

// float32 NCHW <- T NHWC (debug taps)
template <typename T>
int launch_nhwc_to_nchw_f32(const T* x, float* out, int B, int HW, int C, hipStream_t s);
template <typename T>
int launch_nchw_f32_to_nhwc(const float* x, T* out, int B, int HW, int C, hipStream_t s);   // debug entry

}  // namespace prg
