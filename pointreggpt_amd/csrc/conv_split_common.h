// conv_split_common.h — device helpers shared by the f16x3 convolution kernels (conv_split.hip, conv_split512.hip): the on-the-fly
// hi / lo split of eight float32 channels, the fused prologue's SiLU, fragment loads, compile-time tap constants and the scheduling
// pipeline that interleaves a consumer's fragment reads with its MFMAs.
#pragma once
#include "conv_epi.h"
#include "conv_split_ablate.h"

namespace prg {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// 8 consecutive channels -> their hi and lo halves as two 16-byte MFMA operand units
template <bool MIX = true>
__device__ inline void split8(const float (&v)[8], uint4& hi, uint4& lo) {
  f16x8 h, l;
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const f32x2 p = {v[i], v[i + 1]};
    const f16x2 ph = __builtin_convertvector(p, f16x2);          // v_cvt_pk_f16_f32, round to nearest even
    // v - hi, exact in float32.  MIX (round 5): ONE mixed-precision fma per element — v_fma_mix_f32 reads the f16 half directly —
    // instead of a conversion back to float32 and a (packed, two-slot) subtraction: 16 instead of 24 issue slots per eight elements,
    // the same bits.  hipcc folds fma(float(h), -1, v) back into convert + subtract, hence two lines of inline asm (op_sel picks the
    // half).  Same-box A/B (profiles/r05_ab_split8_fma_mix.txt): the persistent 64-channel kernel 248-255 -> 229 us per level-0
    // launch (K = 1152: 460-470 -> 424); the launches with a fused GroupNorm + SiLU prologue got 3-4 % SLOWER with it (the asm
    // statements pin the schedule around the two transcendentals), so those keep the plain form (split8<false>).
    f32x2 r;
    if constexpr (MIX) {
      const uint32_t phw = __builtin_bit_cast(uint32_t, ph);
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(phw), "v"(p[0]));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(phw), "v"(p[1]));
    } else {
      r = p - __builtin_convertvector(ph, f32x2);
    }
    const f16x2 pl = __builtin_convertvector(r, f16x2);
    h[i] = ph[0]; h[i + 1] = ph[1];
    l[i] = pl[0]; l[i + 1] = pl[1];
  }
  hi = __builtin_bit_cast(uint4, h);
  lo = __builtin_bit_cast(uint4, l);
}

// SiLU of the fused prologue: hardware exp2 / reciprocal (1 ulp each) — ~2e-7 relative, the size of the contraction's own error
__device__ inline float silu_fast(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

__device__ inline f16x8 ld_frag(const uint4* p) { return __builtin_bit_cast(f16x8, *p); }

// scheduling pipeline of one k16 step of a consumer wave: (MFMA, ds_read) x 8, then 4 MFMAs  (sched_group_barrier masks: 0x008 = MFMA,
// 0x100 = DS read); placed behind the twelve MFMAs and the eight fragment reads of the OTHER register set it orders
__device__ inline void interleave_8_reads_12_mfmas() {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  }
  __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
}

template <int N>
struct IC {
  static constexpr int value = N;
};

// f(IC<0>()), f(IC<1>()), ... f(IC<N - 1>()): a loop whose index is a compile-time constant in the body
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IC<I>());
    static_for<N, I + 1>(f);
  }
}

// two consecutive channels -> packed hi and lo halves (one pair of split8: the same instructions, the same bits)
template <bool MIX>
__device__ inline void split2(float v0, float v1, uint32_t& hi, uint32_t& lo) {
  const f32x2 p = {v0, v1};
  const f16x2 ph = __builtin_convertvector(p, f16x2);
  f32x2 r;
  if constexpr (MIX) {
    const uint32_t phw = __builtin_bit_cast(uint32_t, ph);
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(phw), "v"(p[0]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(phw), "v"(p[1]));
  } else {
    r = p - __builtin_convertvector(ph, f32x2);
  }
  const f16x2 pl = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(uint32_t, ph);
  lo = __builtin_bit_cast(uint32_t, pl);
}

}  // namespace prg
