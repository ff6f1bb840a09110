// attn_split.hip — Residual(PreNorm(LinearAttention)) (sd:583-589, 631-639, 737-769) for the `f16x3` precision mode: float32
// activations in, float32 out, q / k / v never in HBM.  The structure is attn_fused.hip's (three passes that re-read only x
// and recompute the projections on the matrix pipe), with every contraction as THREE f16 MFMAs on hi/lo-split operands
// (conv_split.hip: a = hi + lo, 22-bit operands) and float32 everywhere else:
//
//   la_ctx   x -> LayerNorm -> k, v;  sweep 1: column maxima of k over the block's slab;  sweep 2 (x again, from L2):
//            p = exp2(k - max) <= 1, sum_n p, ctx[d][e] += p[n][d] v[n][e]         -> per slab (max, sum, ctx)
//   la_fin   slabs merged on their own maxima: ctx = sum_s 2^(m_s - M) ctx_s / sum_s 2^(m_s - M) sum_s / N * 32^-1/2,
//            split into f16 halves in the k-slot order la_out's MFMA wants (float64 sums, fixed slab order)
//   la_out   x -> LayerNorm -> q -> softmax over d (measured maximum) -> out = ctx^T q -> y = Wout out + b -> LayerNorm -> + x
//
// Why not the bf16 kernels' "no shift at all" trick: f16 has no float32 range — exp2(k) with |k| up to 58 overflows it — so
// every exponential is taken relative to a MEASURED maximum (per slab for k, per pixel for q) and is <= 1; what the f16 halves
// then drop is below 2^-22 of the largest term of the sum it enters.
// The unfused float32 path (blocks.hip) moves ~7 GB per level-0 instance at B = 64 (LayerNorm out, the 384-channel qkv tensor
// written once and read three times, ...): 2.0 ms; this file reads x four times and writes it once.
// MFMA: v_mfma_f32_32x32x16_f16, D[i][j] = sum_k A[i][k] B[k][j]; lane l supplies A[l&31][8(l>>5)..+7] and
// B[8(l>>5)..+7][l&31] and receives D[(r&3) + 8(r>>2) + 4(l>>5)][l&31] in register r.
#include <atomic>
#include <cstdlib>

#include "blocks.h"

namespace prg {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) _Float16 h4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kTP = 64;              // pixels per tile
constexpr int kHid = 128;            // heads x dim_head
constexpr int kLdO = kHid + 8;       // LDS row stride of the 128-wide attention-output rows (halves)
constexpr float kLnEps = 1e-5f;

__host__ __device__ inline int sp_tpb(int ntiles) {      // tiles per block: a function of the image size only (batch-independent slabs)
  const int t = ntiles / 8;
  return t < 1 ? 1 : (t > 8 ? 8 : t);
}

__device__ inline f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int e = 0; e < 16; ++e) z[e] = 0.0f;
  return z;
}
__device__ inline float quad_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  return v;
}
// acc += A B with both operands split: the two cross terms first (small), then hi * hi
__device__ inline f32x16 mma3(const h8& ah, const h8& al, const h8& bh, const h8& bl, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
}
// v - hi as one v_fma_mix_f32 (conv_split.hip's split8, where it took 8 % off the persistent 64-channel kernel) bought nothing in
// these kernels (same-box A/B, round 5: la_ctx -1 %, la_out +1-2 %): off; -DPRG_ATTN_SPLIT_MIX=1 builds it
#ifndef PRG_ATTN_SPLIT_MIX
#define PRG_ATTN_SPLIT_MIX 0
#endif
__device__ inline void split1(float v, _Float16& h, _Float16& l) {
  h = (_Float16)v;
#if PRG_ATTN_SPLIT_MIX
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
  l = (_Float16)r;
#else
  l = (_Float16)(v - (float)h);
#endif
}
#define SPLIT_TO(v, H, L, idx)                \
  do {                                        \
    _Float16 sh_, sl_;                        \
    split1((v), sh_, sl_);                    \
    (H)[idx] = sh_;                           \
    (L)[idx] = sl_;                           \
  } while (0)

template <int C>
struct SG {
  static constexpr int LDW = C + 8;       // LDS row stride of C-wide rows (halves): 16 bytes of padding, conflict-free b128 reads
  static constexpr int VPT = C / 16;      // float4 vectors per thread of a 64 x C tile (4 threads per pixel row)
  static constexpr int KK = C / 16;       // MFMA k-steps over C
};

// x tile: global (float32) -> registers -> LayerNorm -> hi / lo f16 tiles in LDS
template <int C>
struct XTileF {
  float4 v[SG<C>::VPT];
  __device__ inline void load(const float* x, int64_t pix0) {
    const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
#pragma unroll
    for (int i = 0; i < SG<C>::VPT; ++i) v[i] = *reinterpret_cast<const float4*>(x + (pix0 + row) * C + (part + 4 * i) * 4);
  }
  // (x - mean) * rstd (biased variance, two-pass on the registers, eps 1e-5: sd:619-628), split, into xh / xl [row][c]
  __device__ inline void normalize_to(_Float16* xh, _Float16* xl) const {
    const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
    float f[SG<C>::VPT][4];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < SG<C>::VPT; ++i) {
      f[i][0] = v[i].x; f[i][1] = v[i].y; f[i][2] = v[i].z; f[i][3] = v[i].w;
      s += (f[i][0] + f[i][1]) + (f[i][2] + f[i][3]);
    }
    s = quad_sum(s);
    const float mean = s * (1.0f / C);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < SG<C>::VPT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f[i][j] -= mean;
        q = fmaf(f[i][j], f[i][j], q);
      }
    q = quad_sum(q);
    const float rstd = 1.0f / __builtin_sqrtf(q * (1.0f / C) + kLnEps);
#pragma unroll
    for (int i = 0; i < SG<C>::VPT; ++i) {
      h4 a, b;
#pragma unroll
      for (int j = 0; j < 4; ++j) SPLIT_TO(f[i][j] * rstd, a, b, j);
      *reinterpret_cast<h4*>(xh + row * SG<C>::LDW + (part + 4 * i) * 4) = a;
      *reinterpret_cast<h4*>(xl + row * SG<C>::LDW + (part + 4 * i) * 4) = b;
    }
  }
};

__device__ inline h8 frag(const _Float16* rows, int ld, int l31, int hi, int kk) {
  return *reinterpret_cast<const h8*>(rows + l31 * ld + kk * 16 + hi * 8);
}
// k-step fragments of rows [r0, r0 + 32) of a row-major [.][COLS] f16 matrix, global -> registers (kept for the whole block)
template <int COLS>
__device__ inline void load_wfrags(h8 (&w)[COLS / 16], const uint16_t* mat, int r0, int l31, int hi) {
#pragma unroll
  for (int kk = 0; kk < COLS / 16; ++kk)
    w[kk] = *reinterpret_cast<const h8*>(mat + (size_t)(r0 + l31) * COLS + kk * 16 + hi * 8);
}

// =====================================================================================================
// la_ctx: per slab (column maxima of k, sums of p = exp2(k - max), p^T v)
// =====================================================================================================
// ONLINE (round 5): ONE sweep over the slab instead of two.  The two-sweep form reads x, LayerNorms it and contracts k TWICE (first
// for the column maxima, then for p = exp2(k - max)); here the maximum of column d is a RUNNING one — a lane owns column d = 32 wave +
// l31 of k and sees a 32-pixel half tile's 16 + 16 values in its own and its partner lane's registers — and whenever a half tile raises
// any column's maximum (a wave-uniform test: after a slab's first tiles it almost never does) the wave rescales what it has
// accumulated so far: row d of ctx and of the row sums times 2^(m_old - m_new), the 32 factors handed from the column-owning lanes to
// the row-owning registers through 128 bytes of wave-private LDS.  Same result up to rounding — every p is still relative to a
// measured maximum and <= 1, the slab leaves (max, sum, ctx) relative to its FINAL maximum, and la_fin merges the slabs as before.
template <int C, bool ONLINE>
__global__ __launch_bounds__(256, C == 64 ? 2 : 1) void la_ctx_split_kernel(const float* __restrict__ x, const uint16_t* __restrict__ wqkv_h,
                                                              const uint16_t* __restrict__ wqkv_l, float* __restrict__ ctxp,
                                                              float* __restrict__ sump, float* __restrict__ maxp, int N, int nslab) {
  using G = SG<C>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  _Float16* xh = reinterpret_cast<_Float16*>(smem);      // [64][LDW]
  _Float16* xl = xh + kTP * G::LDW;
  const int b = blockIdx.y, slab = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  float* const scw = reinterpret_cast<float*>(xl + kTP * G::LDW) + 32 * wave;   // ONLINE: this wave's 32 rescale factors
  const int ntiles = N / kTP;
  const int tpb = sp_tpb(ntiles), t0 = slab * tpb, t1 = min(t0 + tpb, ntiles);
  h8 wkh[G::KK], wkl[G::KK], wvh[G::KK], wvl[G::KK];     // k and v rows of head `wave`
  load_wfrags<C>(wkh, wqkv_h, kHid + 32 * wave, l31, hi);
  load_wfrags<C>(wkl, wqkv_l, kHid + 32 * wave, l31, hi);
  load_wfrags<C>(wvh, wqkv_h, 2 * kHid + 32 * wave, l31, hi);
  load_wfrags<C>(wvl, wqkv_l, 2 * kHid + 32 * wave, l31, hi);
  XTileF<C> xt;
  float m = -INFINITY;                                 // maximum of column d = 32 wave + l31 of k (log2 units: the rows carry log2 e)
  if (t0 < t1) xt.load(x, (int64_t)b * N + (int64_t)t0 * kTP);
  if constexpr (!ONLINE) {
    // ---- sweep 1: the slab's column maxima ----
    for (int t = t0; t < t1; ++t) {
      xt.normalize_to(xh, xl);
      __syncthreads();
      if (t + 1 < t1) xt.load(x, (int64_t)b * N + (int64_t)(t + 1) * kTP);
      else xt.load(x, (int64_t)b * N + (int64_t)t0 * kTP);            // (sweep 2 starts over)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        f32x16 k1 = zero16();
#pragma unroll
        for (int kk = 0; kk < G::KK; ++kk)
          k1 = mma3(frag(xh + pt * 32 * G::LDW, G::LDW, l31, hi, kk), frag(xl + pt * 32 * G::LDW, G::LDW, l31, hi, kk), wkh[kk], wkl[kk], k1);
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, k1[r]);
      }
      __syncthreads();
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
  }
  // ---- the sweep that accumulates ----
  f32x16 ctx = zero16();                               // rows d, column e = l31 of head `wave`
  f32x16 psum = zero16();                              // rows d (any column)
  h8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (_Float16)1.0f;
  for (int t = t0; t < t1; ++t) {
    xt.normalize_to(xh, xl);
    __syncthreads();
    if (t + 1 < t1) xt.load(x, (int64_t)b * N + (int64_t)(t + 1) * kTP);
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      f32x16 k1 = zero16(), v1 = zero16();
#pragma unroll
      for (int kk = 0; kk < G::KK; ++kk) {
        const h8 fh = frag(xh + pt * 32 * G::LDW, G::LDW, l31, hi, kk), fl = frag(xl + pt * 32 * G::LDW, G::LDW, l31, hi, kk);
        k1 = mma3(fh, fl, wkh[kk], wkl[kk], k1);
        v1 = mma3(fh, fl, wvh[kk], wvl[kk], v1);
      }
      if constexpr (ONLINE) {
        float mx = k1[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, k1[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (__builtin_amdgcn_ballot_w64(mx > m) != 0ull) {            // (wave-uniform) some column's maximum rose
          const float mn = fmaxf(m, mx);
          const float sc = __builtin_amdgcn_exp2f(m - mn);            // 1 where it did not; 0 on the slab's first half tile (m = -inf)
          m = mn;
          if (hi == 0) scw[l31] = sc;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {                            // registers 4 g4 .. + 3 are rows d = 8 g4 + 4 hi + {0..3}
            const float4 s4 = *reinterpret_cast<const float4*>(scw + 8 * g4 + 4 * hi);
            ctx[4 * g4] *= s4.x; ctx[4 * g4 + 1] *= s4.y; ctx[4 * g4 + 2] *= s4.z; ctx[4 * g4 + 3] *= s4.w;
            psum[4 * g4] *= s4.x; psum[4 * g4 + 1] *= s4.y; psum[4 * g4 + 2] *= s4.z; psum[4 * g4 + 3] *= s4.w;
          }
          __builtin_amdgcn_wave_barrier();                            // (the factors are read before a later half tile rewrites them)
        }
      }
      // p and v feed the context MFMA straight from the accumulator registers: lane (column, half hi) holds in registers
      // 8 i .. 8 i + 7 exactly the eight pixels that A's row / B's column l31 supplies for k-slots 8 hi .. + 7 of k-step i
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        h8 ph, pl, vh, vl;
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
          SPLIT_TO(__builtin_amdgcn_exp2f(k1[8 * i + s2] - m), ph, pl, s2);
          SPLIT_TO(v1[8 * i + s2], vh, vl, s2);
        }
        ctx = mma3(ph, pl, vh, vl, ctx);
        psum = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, ones, psum, 0, 0, 0);
        psum = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, ones, psum, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  const size_t ph_ = ((size_t)b * 4 + wave) * nslab + slab;
  if (l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sump[ph_ * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = psum[r];
  }
  if (hi == 0) maxp[ph_ * 32 + l31] = m;
#pragma unroll
  for (int r = 0; r < 16; ++r) ctxp[ph_ * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = ctx[r];
}

// =====================================================================================================
// la_fin: slabs merged on their own maxima (float64, fixed slab order) -> ctx^T as f16 halves [b][h][e][slot(d)]
// =====================================================================================================
__global__ __launch_bounds__(256) void la_fin_split_kernel(const float* __restrict__ ctxp, const float* __restrict__ sump,
                                                           const float* __restrict__ maxp, uint16_t* __restrict__ ctxT_h,
                                                           uint16_t* __restrict__ ctxT_l, int N, int nslab) {
  const int h = blockIdx.x, b = blockIdx.y;
  const size_t ph = ((size_t)b * 4 + h) * nslab;
  const double scale = 0.17677669529663687 / (double)N;    // 32^-1/2 (q) and 1/N (v)
  for (int idx = threadIdx.x; idx < 1024; idx += 256) {
    const int d = idx >> 5, e = idx & 31;
    // eight slabs' loads in flight at a time (the serial form waited for every load: 58 us per launch for 33 MB); the sums keep
    // their fixed slab order
    float M = -INFINITY;
    for (int s0 = 0; s0 < nslab; s0 += 8) {
      float m8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) m8[i] = s0 + i < nslab ? maxp[(ph + s0 + i) * 32 + d] : -INFINITY;
#pragma unroll
      for (int i = 0; i < 8; ++i) M = fmaxf(M, m8[i]);
    }
    double c = 0.0, s = 0.0;
    for (int s0 = 0; s0 < nslab; s0 += 8) {
      float m8[8], c8[8], u8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool ok = s0 + i < nslab;
        const size_t q = ph + (ok ? s0 + i : 0);
        m8[i] = ok ? maxp[q * 32 + d] : -INFINITY;
        c8[i] = ctxp[q * 1024 + idx];
        u8[i] = sump[q * 32 + d];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (s0 + i < nslab) {
          const double w = (double)__builtin_amdgcn_exp2f(m8[i] - M);   // (power of two scale; <= 1)
          c += w * (double)c8[i];
          s += w * (double)u8[i];
        }
      }
    }
    const float val = (float)(c / s * scale);
    _Float16 vh, vl;
    split1(val, vh, vl);
    const int slot = (d >> 4) * 16 + ((d >> 2) & 1) * 8 + ((d >> 3) & 1) * 4 + (d & 3);
    const size_t o = ((size_t)b * 4 + h) * 1024 + e * 32 + slot;
    ctxT_h[o] = __builtin_bit_cast(uint16_t, vh);
    ctxT_l[o] = __builtin_bit_cast(uint16_t, vl);
  }
}

// =====================================================================================================
// la_out: q, softmax over d, ctx^T q, to_out conv + bias, LayerNorm, residual
// =====================================================================================================
template <int C>
__global__ __launch_bounds__(256, C == 64 ? 2 : 1) void la_out_split_kernel(const float* __restrict__ x, const uint16_t* __restrict__ wqkv_h,
                                                              const uint16_t* __restrict__ wqkv_l, const uint16_t* __restrict__ wout_h,
                                                              const uint16_t* __restrict__ wout_l, const float* __restrict__ bias,
                                                              const float* __restrict__ out_g, const uint16_t* __restrict__ ctxT_h,
                                                              const uint16_t* __restrict__ ctxT_l, float* __restrict__ out, int N) {
  using G = SG<C>;
  constexpr int RT = C / 32;                 // 32-channel row tiles of y
  static_assert(RT == 2 || RT == 4, "C = 64 or 128");
  constexpr int NA = RT / 2;                 // y accumulators per wave
  constexpr int LDY = C + 4;                 // float row stride of the y tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  _Float16* xh = reinterpret_cast<_Float16*>(smem);      // [64][LDW]
  _Float16* xl = xh + kTP * G::LDW;
  _Float16* oh = xl + kTP * G::LDW;                      // [64][kLdO]  attention output, pixel-major
  _Float16* ol = oh + kTP * kLdO;
  float* lnb = reinterpret_cast<float*>(ol + kTP * kLdO);   // [64][RT][2] LayerNorm partial sums
  float* yt = lnb + kTP * RT * 2;                           // [64][LDY] the normalised y tile
  float* bias_l = yt + kTP * LDY;                           // [C]
  float* outg_l = bias_l + C;                               // [C]
  if (threadIdx.x < C) {
    bias_l[threadIdx.x] = bias[threadIdx.x];
    outg_l[threadIdx.x] = out_g[threadIdx.x];
  }
  const int b = blockIdx.y, slab = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int ntiles = N / kTP;
  const int tpb = sp_tpb(ntiles), t0 = slab * tpb, t1 = min(t0 + tpb, ntiles);
  h8 wqh[G::KK], wql[G::KK];                             // q rows of head `wave`
  load_wfrags<C>(wqh, wqkv_h, 32 * wave, l31, hi);
  load_wfrags<C>(wql, wqkv_l, 32 * wave, l31, hi);
  // ctx^T rows e = l31 of head `wave`; k-slot s of half hi in k-step i is d = 16 i + 8 (s >> 2) + 4 hi + (s & 3): exactly the d
  // of q-accumulator register 8 i + s of a lane in half hi, so q feeds the MFMA without any shuffle
  h8 cah[2], cal[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    cah[i] = *reinterpret_cast<const h8*>(ctxT_h + (((size_t)b * 4 + wave) * 32 + l31) * 32 + i * 16 + hi * 8);
    cal[i] = *reinterpret_cast<const h8*>(ctxT_l + (((size_t)b * 4 + wave) * 32 + l31) * 32 + i * 16 + hi * 8);
  }
  int yrt[NA], ypt[NA];                                  // y accumulators of this wave: row tile / pixel tile
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    if (RT == 2) { yrt[a] = wave & 1; ypt[a] = wave >> 1; }
    else { yrt[a] = wave; ypt[a] = a; }
  }
  h8 woh[kHid / 16], wol[kHid / 16];                     // to_out rows of this wave's y row tile
  load_wfrags<kHid>(woh, wout_h, yrt[0] * 32, l31, hi);
  load_wfrags<kHid>(wol, wout_l, yrt[0] * 32, l31, hi);
  XTileF<C> xt;
  if (t0 < t1) xt.load(x, (int64_t)b * N + (int64_t)t0 * kTP);
  for (int t = t0; t < t1; ++t) {
    const XTileF<C> xraw = xt;                           // residual
    xt.normalize_to(xh, xl);
    __syncthreads();                                                                            // (1) xn ready
    if (t + 1 < t1) xt.load(x, (int64_t)b * N + (int64_t)(t + 1) * kTP);
    // q^T[d][px] of head `wave` (log2 units)
    f32x16 qa[2] = {zero16(), zero16()};
#pragma unroll
    for (int kk = 0; kk < G::KK; ++kk)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
        qa[pt] = mma3(wqh[kk], wql[kk], frag(xh + pt * 32 * G::LDW, G::LDW, l31, hi, kk), frag(xl + pt * 32 * G::LDW, G::LDW, l31, hi, kk), qa[pt]);
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      // softmax over the 32 d of pixel pt * 32 + l31: 16 in this lane, 16 in lane ^ 32
      float mx = qa[pt][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, qa[pt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sm = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        qa[pt][r] = __builtin_amdgcn_exp2f(qa[pt][r] - mx);
        sm += qa[pt][r];
      }
      sm += __shfl_xor(sm, 32, 64);
      const float inv = 1.0f / sm;
      f32x16 oa = zero16();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        h8 qh, ql;
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) SPLIT_TO(qa[pt][8 * i + s2] * inv, qh, ql, s2);
        oa = mma3(cah[i], cal[i], qh, ql, oa);                                   // rows e, column px
      }
      const int px = pt * 32 + l31;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        h4 a, c;
#pragma unroll
        for (int j = 0; j < 4; ++j) SPLIT_TO(oa[4 * g4 + j], a, c, j);
        *reinterpret_cast<h4*>(oh + px * kLdO + 32 * wave + 8 * g4 + 4 * hi) = a;
        *reinterpret_cast<h4*>(ol + px * kLdO + 32 * wave + 8 * g4 + 4 * hi) = c;
      }
    }
    __syncthreads();                                                                            // (2) o tile ready, xn free
    // y^T[c][px] = Wout[c][:] . o[px][:] + bias
    f32x16 ya[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) ya[a] = zero16();
#pragma unroll
    for (int kk = 0; kk < kHid / 16; ++kk)
#pragma unroll
      for (int a = 0; a < NA; ++a)
        ya[a] = mma3(woh[kk], wol[kk], frag(oh + ypt[a] * 32 * kLdO, kLdO, l31, hi, kk), frag(ol + ypt[a] * 32 * kLdO, kLdO, l31, hi, kk), ya[a]);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      float s1 = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = yrt[a] * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        ya[a][r] += bias_l[c];
        s1 += ya[a][r];
      }
      s1 += __shfl_xor(s1, 32, 64);
      if (hi == 0) lnb[((ypt[a] * 32 + l31) * RT + yrt[a]) * 2] = s1;
    }
    __syncthreads();                                                                            // (3) sums
    float mean[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int px = ypt[a] * 32 + l31;
      float s1 = 0.0f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) s1 += lnb[(px * RT + rt) * 2];
      mean[a] = s1 * (1.0f / C);
      float s2 = 0.0f;                                   // two-pass variance, like the reference's x.var()
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float dlt = ya[a][r] - mean[a];
        s2 = fmaf(dlt, dlt, s2);
      }
      s2 += __shfl_xor(s2, 32, 64);
      if (hi == 0) lnb[(px * RT + yrt[a]) * 2 + 1] = s2;
    }
    __syncthreads();                                                                            // (4) squared deviations
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int px = ypt[a] * 32 + l31;
      float s2 = 0.0f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) s2 += lnb[(px * RT + rt) * 2 + 1];
      const float rstd = 1.0f / __builtin_sqrtf(s2 * (1.0f / C) + kLnEps);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int c0 = yrt[a] * 32 + 8 * g4 + 4 * hi;
        const float4 gg = *reinterpret_cast<const float4*>(outg_l + c0);
        float4 w;
        w.x = (ya[a][4 * g4] - mean[a]) * rstd * gg.x;
        w.y = (ya[a][4 * g4 + 1] - mean[a]) * rstd * gg.y;
        w.z = (ya[a][4 * g4 + 2] - mean[a]) * rstd * gg.z;
        w.w = (ya[a][4 * g4 + 3] - mean[a]) * rstd * gg.w;
        *reinterpret_cast<float4*>(yt + px * LDY + c0) = w;
      }
    }
    __syncthreads();                                                                            // (5) y tile ready
    {
      const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
#pragma unroll
      for (int i = 0; i < G::VPT; ++i) {
        const float4 yv = *reinterpret_cast<const float4*>(yt + row * LDY + (part + 4 * i) * 4);
        const float4 xv = xraw.v[i];
        float4 o;
        o.x = yv.x + xv.x; o.y = yv.y + xv.y; o.z = yv.z + xv.z; o.w = yv.w + xv.w;
        *reinterpret_cast<float4*>(out + ((int64_t)b * N + (int64_t)t * kTP + row) * C + (part + 4 * i) * 4) = o;
      }
    }
  }
}

template <int C>
constexpr size_t lds_ctx() { return (size_t)2 * kTP * SG<C>::LDW * 2 + 4 * 32 * sizeof(float); }   // x tile halves + the online form's rescale factors
template <int C>
constexpr size_t lds_out() {
  return (size_t)2 * kTP * SG<C>::LDW * 2 + (size_t)2 * kTP * kLdO * 2 + (size_t)kTP * (C / 32) * 2 * 4 + (size_t)kTP * (C + 4) * 4 + 2 * C * 4;
}

template <int C>
int launch_c(const float* x, const uint16_t* wqkv_h, const uint16_t* wqkv_l, const uint16_t* wout_h, const uint16_t* wout_l,
             const float* bias, const float* out_g, float* out, float* ws, int B, int N, hipStream_t s) {
  static DeviceOnce attr;
  if (!attr.done()) {
    PRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&la_out_split_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_out<C>()));
    attr.mark();
  }
  const int ntiles = N / kTP, nslab = ceil_div(ntiles, sp_tpb(ntiles));
  float* ctxp = ws;
  float* sump = ctxp + (size_t)B * 4 * nslab * 1024;
  float* maxp = sump + (size_t)B * 4 * nslab * 32;
  uint16_t* ctxT_h = reinterpret_cast<uint16_t*>(maxp + (size_t)B * 4 * nslab * 32);
  uint16_t* ctxT_l = ctxT_h + (size_t)B * 4 * 1024;
  const dim3 grid(nslab, B);
  // PRG_SPLIT_LA_ONLINE=0: the two-sweep form of rounds 4 (column maxima first)
  static const int online = [] { const char* e = std::getenv("PRG_SPLIT_LA_ONLINE"); return e ? std::atoi(e) : 1; }();
  if (online) la_ctx_split_kernel<C, true><<<grid, 256, lds_ctx<C>(), s>>>(x, wqkv_h, wqkv_l, ctxp, sump, maxp, N, nslab);
  else la_ctx_split_kernel<C, false><<<grid, 256, lds_ctx<C>(), s>>>(x, wqkv_h, wqkv_l, ctxp, sump, maxp, N, nslab);
  PRG_LAUNCH_CHECK();
  la_fin_split_kernel<<<dim3(4, B), 256, 0, s>>>(ctxp, sump, maxp, ctxT_h, ctxT_l, N, nslab);
  PRG_LAUNCH_CHECK();
  la_out_split_kernel<C><<<grid, 256, lds_out<C>(), s>>>(x, wqkv_h, wqkv_l, wout_h, wout_l, bias, out_g, ctxT_h, ctxT_l, out, N);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}


// =====================================================================================================
// Bottleneck attention core (sd:789-795) for the f16x3 mode (round 5): attn_fused.hip's full_attn_mfma structure on float32
// q / k / v with split-f16 contractions.  One block per (head, image, query share): K (hi / lo) row-major and V^T (hi / lo, keys in
// the order the score accumulators supply them as the B operand) of a block of up to 256 keys in LDS; a lane owns ONE query and sees
// a key tile's 32 scores in its registers.  Two passes over the key blocks — row maxima, then exp / row sums / V^T P — i.e. the
// reference's softmax order; P = exp(s - max) <= 1 is split like every other operand.  The parity mode's scalar kernel
// (full_attn_kernel<float>, float64 sums) took 210 us per evaluation at B = 64, 128 x 128.
// =====================================================================================================
template <int NT, int QS>   // N = 32 NT tokens; QS blocks share a (head, image), each takes every QS-th group of four query tiles
__global__ __launch_bounds__(256) void full_attn_split_kernel(const float* __restrict__ qkv, float* __restrict__ out) {
  constexpr int N = 32 * NT, KB = N < 256 ? N : 256, NKB = N / KB, KT = KB / 32, QPW = NT / (4 * QS);
  constexpr int LDK = 40, LDV = KB + 8;
  static_assert(QPW >= 1 && QPW * 4 * QS == NT, "query tiles per wave");
  extern __shared__ __attribute__((aligned(16))) char smem_fa[];
  _Float16* const Kh = reinterpret_cast<_Float16*>(smem_fa);   // [KB][LDK]
  _Float16* const Kl = Kh + KB * LDK;
  _Float16* const Vh = Kl + KB * LDK;                            // [32][LDV]
  _Float16* const Vl = Vh + 32 * LDV;
  const int h = blockIdx.x, b = blockIdx.y, qs = blockIdx.z, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const float* base = qkv + (size_t)b * N * 384;
  constexpr float kScale = 0.17677669529663687f * 1.4426950408889634f;   // 32^-1/2 log2 e: scores in log2 units

  auto fill = [&](int kb, bool with_v) {
    for (int i = tid; i < KB * 4; i += 256) {
      const int key = i >> 2, u = i & 3;
      const float* kp = base + (size_t)(kb * KB + key) * 384 + 128 + h * 32 + u * 8;
      const float4 a = *reinterpret_cast<const float4*>(kp), c = *reinterpret_cast<const float4*>(kp + 4);
      const float kv[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
      h8 th, tl;
#pragma unroll
      for (int j = 0; j < 8; ++j) SPLIT_TO(kv[j], th, tl, j);
      *reinterpret_cast<h8*>(Kh + key * LDK + u * 8) = th;
      *reinterpret_cast<h8*>(Kl + key * LDK + u * 8) = tl;
      if (with_v) {
        const float4 va = *reinterpret_cast<const float4*>(kp + 128), vc = *reinterpret_cast<const float4*>(kp + 132);
        const float vv[8] = {va.x, va.y, va.z, va.w, vc.x, vc.y, vc.z, vc.w};
        const int d = key & 31;
        const int pos = (key & ~31) + (d >> 4) * 16 + ((d >> 2) & 1) * 8 + ((d >> 3) & 1) * 4 + (d & 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          _Float16 sh_, sl_;
          split1(vv[j], sh_, sl_);
          Vh[(u * 8 + j) * LDV + pos] = sh_;
          Vl[(u * 8 + j) * LDV + pos] = sl_;
        }
      }
    }
  };
  // the lane's query (tile qt, row l31): k-slots 8 hi .. + 7 of the two k-steps, split
  auto load_q = [&](int qt, h8 (&qh)[2], h8 (&ql)[2]) {
    const float* qp = base + (size_t)(qt * 32 + l31) * 384 + h * 32 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float4 a = *reinterpret_cast<const float4*>(qp + ks * 16), c = *reinterpret_cast<const float4*>(qp + ks * 16 + 4);
      const float qv[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) SPLIT_TO(qv[j], qh[ks], ql[ks], j);
    }
  };
  auto scores = [&](int kt, const h8 (&qh)[2], const h8 (&ql)[2]) {
    f32x16 sa = zero16();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const h8 kh = *reinterpret_cast<const h8*>(Kh + (kt * 32 + l31) * LDK + ks * 16 + hi * 8);
      const h8 kl = *reinterpret_cast<const h8*>(Kl + (kt * 32 + l31) * LDK + ks * 16 + hi * 8);
      sa = mma3(kh, kl, qh[ks], ql[ks], sa);
    }
    return sa;
  };

  float m[QPW], l[QPW];
  f32x16 oacc[QPW];
#pragma unroll
  for (int qi = 0; qi < QPW; ++qi) { m[qi] = -INFINITY; l[qi] = 0.0f; oacc[qi] = zero16(); }
  // pass 1: row maxima
  for (int kb = 0; kb < NKB; ++kb) {
    if (kb > 0) __syncthreads();
    fill(kb, NKB == 1);
    __syncthreads();
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
      h8 qh[2], ql[2];
      load_q((qs * QPW + qi) * 4 + wave, qh, ql);
#pragma unroll 2
      for (int kt = 0; kt < KT; ++kt) {
        const f32x16 sa = scores(kt, qh, ql);
#pragma unroll
        for (int r = 0; r < 16; ++r) m[qi] = fmaxf(m[qi], sa[r] * kScale);
      }
    }
  }
#pragma unroll
  for (int qi = 0; qi < QPW; ++qi) m[qi] = fmaxf(m[qi], __shfl_xor(m[qi], 32, 64));
  // pass 2: p = 2^(s - max), row sums, O^T += V^T P
  for (int kb = 0; kb < NKB; ++kb) {
    if (NKB > 1) {
      __syncthreads();
      fill(kb, true);
      __syncthreads();
    }
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
      h8 qh[2], ql[2];
      load_q((qs * QPW + qi) * 4 + wave, qh, ql);
#pragma unroll 2
      for (int kt = 0; kt < KT; ++kt) {
        const f32x16 sa = scores(kt, qh, ql);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          h8 ph, pl;
#pragma unroll
          for (int s2 = 0; s2 < 8; ++s2) {
            const float pv = __builtin_amdgcn_exp2f(fmaf(sa[8 * i + s2], kScale, -m[qi]));
            l[qi] += pv;
            SPLIT_TO(pv, ph, pl, s2);
          }
          const h8 vh = *reinterpret_cast<const h8*>(Vh + l31 * LDV + kt * 32 + i * 16 + hi * 8);
          const h8 vl = *reinterpret_cast<const h8*>(Vl + l31 * LDV + kt * 32 + i * 16 + hi * 8);
          oacc[qi] = mma3(vh, vl, ph, pl, oacc[qi]);
        }
      }
    }
  }
#pragma unroll
  for (int qi = 0; qi < QPW; ++qi) {
    const float lt = l[qi] + __shfl_xor(l[qi], 32, 64);
    const float inv = 1.0f / lt;
    const int qt = (qs * QPW + qi) * 4 + wave;
    float* op = out + ((size_t)b * N + qt * 32 + l31) * 128 + h * 32 + 4 * hi;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      *reinterpret_cast<float4*>(op + 8 * g4) = make_float4(oacc[qi][4 * g4] * inv, oacc[qi][4 * g4 + 1] * inv, oacc[qi][4 * g4 + 2] * inv,
                                                            oacc[qi][4 * g4 + 3] * inv);
  }
}

template <int NT, int QS>
int launch_fa(const float* qkv, float* out, int B, hipStream_t s) {
  constexpr int N = 32 * NT, KB = N < 256 ? N : 256;
  constexpr size_t lds = (size_t)(2 * KB * 40 + 2 * 32 * (KB + 8)) * 2;
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    PRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(full_attn_split_kernel<NT, QS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark();
  }
  full_attn_split_kernel<NT, QS><<<dim3(4, B, QS), 256, lds, s>>>(qkv, out);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

}  // namespace

bool linattn_split_supported(int C, int N) {
  static const int c128 = [] { const char* e = std::getenv("PRG_SPLIT_ATTN_C128"); return e ? std::atoi(e) : 1; }();
  return (C == 64 || (C == 128 && c128)) && N % kTP == 0 && N >= kTP;
}

size_t linattn_split_ws_floats(int B, int N) {
  const int ntiles = N / kTP, nslab = ceil_div(ntiles, sp_tpb(ntiles));
  return (size_t)B * 4 * nslab * (1024 + 64) + (size_t)B * 4 * 1024 + 64;   // slabs (ctx, sum, max) + ctx^T halves (2 x f16 = 1 float each)
}

// x, out: (B, N, C) float32.  wqkv_h / _l: f16 halves of [384][C] (PreNorm gain folded in, q and k rows times log2 e);
// wout_h / _l: [C][128]; bias, out_g: [C] float32.
int launch_linear_attention_split(const float* x, const uint16_t* wqkv_h, const uint16_t* wqkv_l, const uint16_t* wout_h,
                                  const uint16_t* wout_l, const float* bias, const float* out_g, float* out, float* ws, int B, int N,
                                  int C, hipStream_t s) {
  PRG_CHECK(linattn_split_supported(C, N) && x && out && ws, "linear attention (f16x3): unsupported shape");
  if (C == 128) return launch_c<128>(x, wqkv_h, wqkv_l, wout_h, wout_l, bias, out_g, out, ws, B, N, s);
  return launch_c<64>(x, wqkv_h, wqkv_l, wout_h, wout_l, bias, out_g, out, ws, B, N, s);
}

// Bottleneck attention core on split-f16 MFMAs: qkv (B, N, 384) float32 -> out (B, N, 128) float32.  PRG_SPLIT_FULLATTN=0: never.
bool full_attention_split_supported(int N) {
  static const int on = [] { const char* e = std::getenv("PRG_SPLIT_FULLATTN"); return e ? std::atoi(e) : 1; }();
  return on && (N == 128 || N == 256 || N == 512 || N == 1024);
}

int launch_full_attention_split(const float* qkv, float* out, int B, int N, hipStream_t s) {
  PRG_CHECK(full_attention_split_supported(N) && qkv && out && B > 0, "attention (f16x3): unsupported shape");
  // query shares so that a small batch still covers the chip (B x 4 heads x QS blocks)
  if (N == 128) return launch_fa<4, 1>(qkv, out, B, s);
  if (N == 256) return B >= 32 ? launch_fa<8, 1>(qkv, out, B, s) : launch_fa<8, 2>(qkv, out, B, s);
  if (N == 512) return B >= 32 ? launch_fa<16, 1>(qkv, out, B, s) : launch_fa<16, 4>(qkv, out, B, s);
  return B >= 32 ? launch_fa<32, 2>(qkv, out, B, s) : launch_fa<32, 4>(qkv, out, B, s);
}

}  // namespace prg
