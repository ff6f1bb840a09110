// attn_fused.hip — Residual(PreNorm(LinearAttention)) of the U-Net (sd:583-589, 631-639, 737-769) as four kernels that
// never materialise q, k, v for the bf16 throughput path.
//
// The unfused path (blocks.hip) moves ~3 GB per instance at 128x128 / B=64 (LayerNorm out, the 384-channel qkv tensor
// written once and read three times, the 128-channel attention output, the to_out result): it is HBM-bound at 8x the
// bytes the block needs.  Here every pass re-reads only x (C channels per pixel) and recomputes what it needs with MFMA:
//
//   la_kmax   x -> LayerNorm -> k = Wk x^        column maxima per slab   (only when the static bound on |k| does not hold)
//   la_ctx    x -> LayerNorm -> k, v             p = exp(k [- max]);  sum_n p,  ctx[d][e] += p[n][d] v[n][e]   per slab
//   la_fin    slabs summed in fixed order, ctx / sum / N * 32^-1/2 -> bf16, stored in the k-slot order la_out's MFMA wants
//   la_out    x -> LayerNorm -> q -> softmax_d -> out = ctx^T q -> y = Wout out + b -> LayerNorm -> + x     -> store
//
// Tiles are 64 pixels; a block of four waves walks kTilesPerBlock tiles of one image.  Wave w owns head w and keeps its
// rows of the projection weights in registers as MFMA fragments (no LDS copy: 29-46 KB of LDS per block, several
// blocks per CU); the LDS tiles have rows padded to C+8 / 136 / 72 bf16 so the 16 lanes of a ds_read_b128 group hit
// distinct banks.
// MFMA: v_mfma_f32_32x32x16_bf16, D[i][j] = sum_k A[i][k] B[k][j]; lane l supplies A[l&31][8(l>>5)..+7] and
// B[8(l>>5)..+7][l&31] and receives D[(r&3) + 8(r>>2) + 4(l>>5)][l&31] in register r.
// The LayerNorm gain of PreNorm is folded into the projection weights on the host (unet.hip).
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "blocks.h"

namespace prg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

namespace {

constexpr int kTP = 64;              // pixels per tile
constexpr int kHid = 128;            // heads x dim_head
constexpr int kTilesPerBlock = 8;
// Linear attention: tiles per block as a function of the image size ONLY (a batch-independent slab structure keeps a
// scene's result identical for every batch size): 8 from 64 x 64 pixels up, fewer for the small levels, whose launches
// otherwise fill a quarter of the CUs with blocks that walk four tiles in series (16 x 16: 64 blocks -> 256).
// (compile-time overrides for A/B builds; measured round 3: 16 or 32 tiles per la_ctx block 73-75 against 72 us, 4 or 16 per
// la_out block 105-108 against 99 us)
#ifndef PRG_LA_CTX_TPB
#define PRG_LA_CTX_TPB 8
#endif
#ifndef PRG_LA_OUT_TPB
#define PRG_LA_OUT_TPB 8
#endif
__host__ __device__ inline int la_tpb(int ntiles) {             // la_kmax / la_ctx (the slab structure of the partials)
  const int t = ntiles / 8;
  return t < 1 ? 1 : (t > PRG_LA_CTX_TPB ? PRG_LA_CTX_TPB : t);
}
__host__ __device__ inline int la_out_tpb(int ntiles) {         // la_out (pixels are independent there: any partition)
  const int t = ntiles / 8;
  return t < 1 ? 1 : (t > PRG_LA_OUT_TPB ? PRG_LA_OUT_TPB : t);
}
constexpr int kLdO = kHid + 8;       // LDS row stride of 128-wide rows (bf16 elements)
constexpr float kLnEps = 1e-5f;

__device__ inline uint32_t pack2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ inline float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ inline float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ inline float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

template <int C>
struct Geo {
  static constexpr int LDW = C + 8;       // LDS row stride of C-wide rows (bf16 elements)
  static constexpr int VPT = C / 32;      // 16-byte vectors per thread of a 64 x C tile (4 threads per pixel row)
  static constexpr int KK = C / 16;       // MFMA k-steps over C
};

// sum over the four lanes of a quad (quad_perm [1,0,3,2] then [2,3,0,1]), every lane gets the total
__device__ inline float quad_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  return v;
}

// ---- x tile: global -> registers -> LayerNorm -> LDS ------------------------------------------------------------
// thread t: pixel row t >> 2, vectors (t & 3) + 4 i (interleaved so that a row's four threads read one contiguous run)
template <int C>
struct XTile {
  uint4 v[Geo<C>::VPT];
  __device__ inline void load(const bf16_t* x, int64_t pix0, int valid) {
    const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
#pragma unroll
    for (int i = 0; i < Geo<C>::VPT; ++i)
      v[i] = row < valid ? *reinterpret_cast<const uint4*>(x + (pix0 + row) * C + (part + 4 * i) * 8) : make_uint4(0, 0, 0, 0);
  }
  // (x - mean) * rstd as bf16 into xn[row][c]  (biased variance, two-pass on the registers, eps 1e-5)
  __device__ inline void normalize_to(__bf16* xn) const {
    const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
    float f[Geo<C>::VPT][8];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < Geo<C>::VPT; ++i) {
      const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f[i][2 * j] = bf_lo(w[j]);
        f[i][2 * j + 1] = bf_hi(w[j]);
        s += f[i][2 * j] + f[i][2 * j + 1];
      }
    }
    s = quad_sum(s);                                     // a row's four threads are one quad: DPP, no LDS crossbar
    const float mean = s * (1.0f / C);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < Geo<C>::VPT; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f[i][j] -= mean;
        q = fmaf(f[i][j], f[i][j], q);
      }
    q = quad_sum(q);
    const float rstd = __builtin_amdgcn_rsqf(q * (1.0f / C) + kLnEps);   // (v_rsq_f32, 1 ulp: these tiles are rounded to bf16 next)
#pragma unroll
    for (int i = 0; i < Geo<C>::VPT; ++i) {
      uint4 o;
      o.x = pack2(f[i][0] * rstd, f[i][1] * rstd);
      o.y = pack2(f[i][2] * rstd, f[i][3] * rstd);
      o.z = pack2(f[i][4] * rstd, f[i][5] * rstd);
      o.w = pack2(f[i][6] * rstd, f[i][7] * rstd);
      *reinterpret_cast<uint4*>(xn + row * Geo<C>::LDW + (part + 4 * i) * 8) = o;
    }
  }
};

// rows [r0, r0 + nrows) of a row-major [.][C] bf16 matrix -> LDS rows of stride LD (all 256 threads)
template <int COLS, int LD>
__device__ inline void stage_rows(__bf16* dst, const bf16_t* src, int nrows) {
  constexpr int VR = COLS / 8;
  for (int i = threadIdx.x; i < nrows * VR; i += 256) {
    const int r = i / VR, v = i - r * VR;
    *reinterpret_cast<uint4*>(dst + r * LD + v * 8) = *reinterpret_cast<const uint4*>(src + (size_t)r * COLS + v * 8);
  }
}

__device__ inline bf16x8 frag(const __bf16* rows, int ld, int l31, int hi, int kk) {
  return *reinterpret_cast<const bf16x8*>(rows + l31 * ld + kk * 16 + hi * 8);
}

// k-step fragments of 32 rows [r0, r0 + 32) of a row-major [.][COLS] bf16 matrix, straight from global memory into
// registers (each wave keeps its own rows for the whole block: no LDS copy, so several blocks fit on a CU)
template <int COLS>
__device__ inline void load_wfrags(bf16x8 (&w)[COLS / 16], const bf16_t* mat, int r0, int l31, int hi) {
#pragma unroll
  for (int kk = 0; kk < COLS / 16; ++kk)
    w[kk] = *reinterpret_cast<const bf16x8*>(mat + (size_t)(r0 + l31) * COLS + kk * 16 + hi * 8);
}

__device__ inline f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int e = 0; e < 16; ++e) z[e] = 0.0f;
  return z;
}

// =====================================================================================================
// pass 1: column maxima of k per slab
// =====================================================================================================
template <int C>
__global__ __launch_bounds__(256) void la_kmax_fused_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wqkv,
                                                            float* __restrict__ pmax, int N, int nslab) {
  using G = Geo<C>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* xn = reinterpret_cast<__bf16*>(smem);      // [64][LDW]
  const int b = blockIdx.y, slab = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int ntiles = (N + kTP - 1) / kTP;
  const int tpb = la_tpb(ntiles), t0 = slab * tpb, t1 = min(t0 + tpb, ntiles);
  bf16x8 wk[G::KK];                                    // k rows 32 wave .. + 32 of the projection
  load_wfrags<C>(wk, wqkv, kHid + 32 * wave, l31, hi);
  float m = -INFINITY;                                 // column 32 wave + l31, this lane's pixel rows
  XTile<C> xt;
  if (t0 < t1) xt.load(x, (int64_t)b * N + (int64_t)t0 * kTP, min(kTP, N - t0 * kTP));
  for (int t = t0; t < t1; ++t) {
    xt.normalize_to(xn);
    __syncthreads();
    if (t + 1 < t1) xt.load(x, (int64_t)b * N + (int64_t)(t + 1) * kTP, min(kTP, N - (t + 1) * kTP));
    f32x16 acc[2] = {zero16(), zero16()};
#pragma unroll
    for (int kk = 0; kk < G::KK; ++kk) {
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(xn + pt * 32 * G::LDW, G::LDW, l31, hi, kk), wk[kk], acc[pt], 0, 0, 0);
    }
    const int valid = min(kTP, N - t * kTP);
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int px = pt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (px < valid) m = fmaxf(m, acc[pt][r]);
      }
    __syncthreads();
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  if (hi == 0) pmax[((size_t)b * nslab + slab) * kHid + 32 * wave + l31] = m;
}

// =====================================================================================================
// pass 2: per-slab sums of p = exp(k - max) and of p v^T
// =====================================================================================================
// Both kernels of this file's attention are VALU-ISSUE bound (SQ counters, round 3: per tile and wave ~330 VALU instructions
// beside 24 MFMAs, VALU issue + matrix-pipe busy cycles ~ the kernel's duration, i.e. they barely overlap; neither occupancy
// 3 -> 4, nor the slab count, nor a deeper x prefetch moved it), so the loop is written to issue as few VALU instructions
// as possible:
//  * p and v go into the context MFMA STRAIGHT FROM THE ACCUMULATOR REGISTERS — the contraction index (pixels) may be
//    enumerated in any order as long as both operands use the same one, and lane (column, half hi) of the k / v
//    accumulators holds, in registers 8 i .. 8 i + 7, exactly the eight pixels that A's row / B's column `l31` supplies for
//    k-slots 8 hi .. 8 hi + 7 of k-step i (la_out's trick for q): no transposed LDS tiles;
//  * KSTAT (static bound on |k|, <= 57.7 in the log2 units the pre-scaled rows produce — unet.hip): NO shift at all.  The
//    softmax is shift-invariant, exp2(k) stays inside [2^-58, 2^58], the sums over at most 2^16 pixels inside 2^74, and
//    bf16 / fp32 relative precision does not depend on the scale: no splat, no subtraction, no maximum pass.  Otherwise
//    (measured maxima) the shift is the first MFMA's C operand, a loop-invariant register set;
//  * whole tiles take a branch without the per-pixel masks;
//  * PSUM_MFMA: sum_n p from the matrix pipe (pT times an all-ones operand: the SAME bf16-rounded p that feeds ctx) instead
//    of 32 float additions per tile and lane.
template <int C, bool PSUM_MFMA, bool KSTAT>
__global__ __launch_bounds__(256, C == 64 ? 3 : (C == 128 ? 2 : 1)) void la_ctx_fused_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wqkv,
                                                           const float* __restrict__ pmax, float* __restrict__ ctxp,
                                                           float* __restrict__ sump, int N, int nslab) {
  using G = Geo<C>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* xn = reinterpret_cast<__bf16*>(smem);      // [64][LDW]
  const int b = blockIdx.y, slab = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int ntiles = (N + kTP - 1) / kTP;
  const int tpb = la_tpb(ntiles), t0 = slab * tpb, t1 = min(t0 + tpb, ntiles);
  bf16x8 wk[G::KK], wv[G::KK];                         // k and v rows of head `wave`
  load_wfrags<C>(wk, wqkv, kHid + 32 * wave, l31, hi);
  load_wfrags<C>(wv, wqkv, 2 * kHid + 32 * wave, l31, hi);
  f32x16 kinit0 = zero16();                            // the k accumulators' initial value: minus the column maximum
  if constexpr (!KSTAT) {
    float m = -INFINITY;                               // (fixed-order reduce of la_kmax's slab maxima)
    for (int s2 = 0; s2 < nslab; ++s2) m = fmaxf(m, pmax[((size_t)b * nslab + s2) * kHid + 32 * wave + l31]);
#pragma unroll
    for (int e = 0; e < 16; ++e) kinit0[e] = -m;
  }
  f32x16 ctx = zero16();                               // rows d, column e = l31 of head `wave`
  f32x16 psum = zero16();                              // rows d (any column)
  float ssum = 0.0f;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  XTile<C> xt;
  if (t0 < t1) xt.load(x, (int64_t)b * N + (int64_t)t0 * kTP, min(kTP, N - t0 * kTP));
  for (int t = t0; t < t1; ++t) {
    xt.normalize_to(xn);
    __syncthreads();
    if (t + 1 < t1) xt.load(x, (int64_t)b * N + (int64_t)(t + 1) * kTP, min(kTP, N - (t + 1) * kTP));
    const int valid = min(kTP, N - t * kTP);
    // one 32-pixel half: k = x Wk^T (- shift) and v = x Wv^T, p = exp2(k), ctx += p^T v
    auto half = [&](const int pt, auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
      f32x16 k1, v1;
#pragma unroll
      for (int kk = 0; kk < G::KK; ++kk) {
        const bf16x8 xf = frag(xn + pt * 32 * G::LDW, G::LDW, l31, hi, kk);
        k1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, wk[kk], kk == 0 ? (KSTAT ? zero16() : kinit0) : k1, 0, 0, 0);
        v1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, wv[kk], kk == 0 ? zero16() : v1, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        bf16x8 pa, vb;
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
          const int r = 8 * i + s2;
          float pe = __builtin_amdgcn_exp2f(k1[r]);
          if constexpr (!FULL) pe = (pt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) < valid ? pe : 0.0f;
          if constexpr (!PSUM_MFMA) ssum += pe;
          pa[s2] = (__bf16)pe;
          vb[s2] = (__bf16)v1[r];
        }
        ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, vb, ctx, 0, 0, 0);
        if constexpr (PSUM_MFMA) psum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, ones, psum, 0, 0, 0);
      }
    };
    if (valid == kTP) {                                // (wave-uniform)
      half(0, std::true_type{});
      half(1, std::true_type{});
    } else {
      half(0, std::false_type{});
      half(1, std::false_type{});
    }
    __syncthreads();
  }
  const size_t ph = ((size_t)b * 4 + wave) * nslab + slab;
  if constexpr (PSUM_MFMA) {
    if (l31 == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sump[ph * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = psum[r];
    }
  } else {
    ssum += __shfl_xor(ssum, 32, 64);
    if (hi == 0) sump[ph * 32 + l31] = ssum;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) ctxp[ph * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = ctx[r];
}

// =====================================================================================================
// pass 3: ctx = sum_slabs(ctxp) / sum_slabs(sump) / N * 32^-1/2, as bf16 [b][h][e][slot(d)]
// slot(d) orders the 32 d of a row the way la_out's q registers supply them to the MFMA (see there)
// =====================================================================================================
__global__ __launch_bounds__(256) void la_fin_fused_kernel(const float* __restrict__ ctxp, const float* __restrict__ sump,
                                                           bf16_t* __restrict__ ctxT, int N, int nslab) {
  const int h = blockIdx.x, b = blockIdx.y;
  const size_t ph = ((size_t)b * 4 + h) * nslab;
  const float scale = 0.17677669529663687f / (float)N;    // 32^-1/2 (q) and 1/N (v)
  for (int idx = threadIdx.x; idx < 1024; idx += 256) {
    const int d = idx >> 5, e = idx & 31;
    float c = 0.0f, s = 0.0f;
    int s2 = 0;
    for (; s2 + 8 <= nslab; s2 += 8) {          // loads of a group issued together; the sums stay in slab order
      float cv[8], sv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        cv[u] = ctxp[(ph + s2 + u) * 1024 + idx];
        sv[u] = sump[(ph + s2 + u) * 32 + d];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c += cv[u];
        s += sv[u];
      }
    }
    for (; s2 < nslab; ++s2) {
      c += ctxp[(ph + s2) * 1024 + idx];
      s += sump[(ph + s2) * 32 + d];
    }
    const int slot = (d >> 4) * 16 + ((d >> 2) & 1) * 8 + ((d >> 3) & 1) * 4 + (d & 3);
    ctxT[((size_t)b * 4 + h) * 1024 + e * 32 + slot] = f32_to_bf16(c / s * scale);
  }
}

// =====================================================================================================
// pass 4: q, softmax over d, ctx^T q, to_out conv + bias, LayerNorm, residual
// =====================================================================================================
template <int C, bool QSTAT>   // QSTAT: static bound on |q| (qshift != null is the flag; the value itself is not needed)
__global__ __launch_bounds__(256, C == 64 ? 3 : (C == 128 ? 2 : 1)) void la_out_fused_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wqkv,
                                                           const bf16_t* __restrict__ wout, const float* __restrict__ bias,
                                                           const float* __restrict__ out_g, const bf16_t* __restrict__ ctxT,
                                                           bf16_t* __restrict__ out, int N, const float* __restrict__ qshift) {
  using G = Geo<C>;
  constexpr int RT = C / 32;                 // 32-channel row tiles of y
  constexpr int NA = RT / 2;                 // y accumulators per wave (RT x 2 pixel tiles over 4 waves)
  constexpr int NR = RT > 4 ? RT / 4 : 1;    // distinct row tiles per wave (C = 256: two, each with both pixel tiles)
  constexpr int RTL = RT > 4 ? RT : 4;       // row-tile pitch of the LayerNorm partial sums
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* xn = reinterpret_cast<__bf16*>(smem);      // [64][LDW]
  __bf16* ot = xn + kTP * G::LDW;                    // [64][kLdO]  attention output, pixel-major
  float* lnb = reinterpret_cast<float*>(ot + kTP * kLdO);   // [64][4][2] LayerNorm partial sums
  __bf16* yt = reinterpret_cast<__bf16*>(lnb + kTP * RTL * 2);   // [64][LDW] the normalised y tile (its own buffer: no barrier
                                                             // between a tile's last phase and the next tile's LayerNorm)
  // to_out's bias and the output LayerNorm's gain live in LDS: as global loads inside the tile loop they sat BEHIND the
  // next tile's x prefetch in the in-order vmcnt queue, so every tile waited for its successor's HBM latency
  float* bias_l = reinterpret_cast<float*>(yt + kTP * G::LDW);   // [C]
  float* outg_l = bias_l + C;                                    // [C]
  if (threadIdx.x < C) {
    bias_l[threadIdx.x] = bias[threadIdx.x];
    outg_l[threadIdx.x] = out_g[threadIdx.x];
  }                                                              // (visible after the first tile's barrier (1))
  const int b = blockIdx.y, slab = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int ntiles = (N + kTP - 1) / kTP;
  const int tpb = la_out_tpb(ntiles), t0 = slab * tpb, t1 = min(t0 + tpb, ntiles);
  bf16x8 wq[G::KK];                                    // q rows of head `wave`
  load_wfrags<C>(wq, wqkv, 32 * wave, l31, hi);
  // ctx^T rows e = l31 of head `wave`; k-slot s of half hi in k-step i is d = 16 i + 8 (s >> 2) + 4 hi + (s & 3):
  // exactly the d of q-accumulator register 8 i + s of a lane in half hi, so q feeds the MFMA without any shuffle
  bf16x8 ca[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
    ca[i] = *reinterpret_cast<const bf16x8*>(ctxT + (((size_t)b * 4 + wave) * 32 + l31) * 32 + i * 16 + hi * 8);
  // y accumulators of this wave: row tile / pixel tile
  int yrt[NA], ypt[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    if (RT == 2) { yrt[a] = wave & 1; ypt[a] = wave >> 1; }
    else if (RT <= 4) { yrt[a] = wave; ypt[a] = a; }
    else { yrt[a] = wave * NR + (a >> 1); ypt[a] = a & 1; }
  }
  bf16x8 wo[NR][kHid / 16];                            // to_out rows of this wave's y row tile(s)
#pragma unroll
  for (int r = 0; r < NR; ++r) load_wfrags<kHid>(wo[r], wout, yrt[RT > 4 ? 2 * r : 0] * 32, l31, hi);
  XTile<C> xt;
  if (t0 < t1) xt.load(x, (int64_t)b * N + (int64_t)t0 * kTP, min(kTP, N - t0 * kTP));
  for (int t = t0; t < t1; ++t) {
    asm volatile("" ::: "memory");                     // (bias / out_g stay loads inside the loop: hoisted they cost 32 registers and a block per CU)
    const XTile<C> xraw = xt;                          // residual
    const int valid = min(kTP, N - t * kTP);
    xt.normalize_to(xn);
    __syncthreads();                                                                            // (1) xn ready
    if (t + 1 < t1) xt.load(x, (int64_t)b * N + (int64_t)(t + 1) * kTP, min(kTP, N - (t + 1) * kTP));
    // q^T[d][px] of head `wave`.  QSTAT (static bound |q| <= 57.7, see la_ctx's KSTAT): the softmax over d is
    // shift-invariant and exp2(q) stays inside [2^-58, 2^58], so there is no maximum, no shift and no subtraction.
    constexpr bool qstat = QSTAT;
    f32x16 qa[2];
#pragma unroll
    for (int kk = 0; kk < G::KK; ++kk) {
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
        qa[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[kk], frag(xn + pt * 32 * G::LDW, G::LDW, l31, hi, kk),
                                                         kk == 0 ? zero16() : qa[pt], 0, 0, 0);
    }
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      f32x16 oa = zero16();
      if constexpr (qstat) {
        // e^q unnormalised into the contraction; sum_d e^q of pixel (column) l31 from the same operands against an
        // all-ones A (every row of `sa` holds it); the 1 / sum scales the 16 OUTPUTS of the lane instead of its 32 inputs
        f32x16 sa = zero16();
        bf16x8 ones;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          bf16x8 qb;
#pragma unroll
          for (int s2 = 0; s2 < 8; ++s2) qb[s2] = (__bf16)__builtin_amdgcn_exp2f(qa[pt][8 * i + s2]);
          oa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[i], qb, oa, 0, 0, 0);      // rows e, column px
          sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, qb, sa, 0, 0, 0);
        }
        const float inv = __builtin_amdgcn_rcpf(sa[0]);
        const f32x2 inv2 = {inv, inv};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {                  // (v_pk_mul_f32)
          const f32x2 o2 = f32x2{oa[r], oa[r + 1]} * inv2;
          oa[r] = o2[0];
          oa[r + 1] = o2[1];
        }
      } else {
        // measured maximum (blocks whose static bound is too large): softmax over the 32 d of pixel pt*32 + l31, 16 in this
        // lane, 16 in lane ^ 32
        float mx = qa[pt][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, qa[pt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sm = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          qa[pt][r] = __builtin_amdgcn_exp2f(qa[pt][r] - mx);   // q arrives times log2(e) (weights pre-scaled)
          sm += qa[pt][r];
        }
        sm += __shfl_xor(sm, 32, 64);
        const float inv = 1.0f / sm;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          bf16x8 qb;
#pragma unroll
          for (int s2 = 0; s2 < 8; ++s2) qb[s2] = (__bf16)(qa[pt][8 * i + s2] * inv);
          oa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[i], qb, oa, 0, 0, 0);      // rows e, column px
        }
      }
      const int px = pt * 32 + l31;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 w;
        w.x = pack2(oa[4 * g4], oa[4 * g4 + 1]);
        w.y = pack2(oa[4 * g4 + 2], oa[4 * g4 + 3]);
        *reinterpret_cast<uint2*>(ot + px * kLdO + 32 * wave + 8 * g4 + 4 * hi) = w;
      }
    }
    __syncthreads();                                                                            // (2) ot ready, xn free
    // y^T[c][px] = Wout[c][:] . o[px][:] + bias
    f32x16 ya[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) ya[a] = zero16();
#pragma unroll
    for (int kk = 0; kk < kHid / 16; ++kk) {
#pragma unroll
      for (int a = 0; a < NA; ++a)
        ya[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wo[RT > 4 ? (a >> 1) : 0][kk],
                                                        frag(ot + ypt[a] * 32 * kLdO, kLdO, l31, hi, kk), ya[a], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = yrt[a] * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        ya[a][r] += bias_l[c];
        s1 += ya[a][r];
        s2 = fmaf(ya[a][r], ya[a][r], s2);
      }
      // v_permlane32_swap(s1, s2): lanes 0..31 end up with (s1 lower, s1 upper), lanes 32..63 with (s2 lower, s2 upper), so
      // one swap and one add leave the pixel's sum in the lower half and its sum of squares in the upper half
      const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s2), false, false);
      const float tot = __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
      lnb[((ypt[a] * 32 + l31) * RTL + yrt[a]) * 2 + hi] = tot;
    }
    __syncthreads();                                                                            // (3) partial sums
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int px = ypt[a] * 32 + l31;
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        s1 += lnb[(px * RTL + rt) * 2];
        s2 += lnb[(px * RTL + rt) * 2 + 1];
      }
      const float mean = s1 * (1.0f / C);
      const float rstd = __builtin_amdgcn_rsqf(fmaxf(s2 * (1.0f / C) - mean * mean, 0.0f) + kLnEps);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int c0 = yrt[a] * 32 + 8 * g4 + 4 * hi;
        const float4 gg = *reinterpret_cast<const float4*>(outg_l + c0);
        uint2 w;
        w.x = pack2((ya[a][4 * g4] - mean) * rstd * gg.x, (ya[a][4 * g4 + 1] - mean) * rstd * gg.y);
        w.y = pack2((ya[a][4 * g4 + 2] - mean) * rstd * gg.z, (ya[a][4 * g4 + 3] - mean) * rstd * gg.w);
        *reinterpret_cast<uint2*>(yt + px * G::LDW + c0) = w;
      }
    }
    __syncthreads();                                                                            // (4) y tile ready
    {
      const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
      if (row < valid) {
#pragma unroll
        for (int i = 0; i < G::VPT; ++i) {
          const uint4 yv = *reinterpret_cast<const uint4*>(yt + row * G::LDW + (part + 4 * i) * 8);
          const uint4 xv = xraw.v[i];
          uint4 o;
          o.x = pack2(bf_lo(yv.x) + bf_lo(xv.x), bf_hi(yv.x) + bf_hi(xv.x));
          o.y = pack2(bf_lo(yv.y) + bf_lo(xv.y), bf_hi(yv.y) + bf_hi(xv.y));
          o.z = pack2(bf_lo(yv.z) + bf_lo(xv.z), bf_hi(yv.z) + bf_hi(xv.z));
          o.w = pack2(bf_lo(yv.w) + bf_lo(xv.w), bf_hi(yv.w) + bf_hi(xv.w));
          *reinterpret_cast<uint4*>(out + ((int64_t)b * N + (int64_t)t * kTP + row) * C + (part + 4 * i) * 8) = o;
        }
      }
    }
  }
}

// =====================================================================================================
// Bottleneck attention core (sd:789-795) on MFMA for N <= 256 tokens: one block per (image, head).
//   S^T[key][query] = K q^T (keys as MFMA rows: a lane owns ONE query and sees all its keys in registers, so the
//   softmax over keys is register-local plus one lane^32 exchange), P = exp(S/sqrt(32) - max), O^T = V^T P.
// K goes to LDS row-major, V transposed with the keys of each 32-key tile stored in the order the S^T accumulator
// registers supply them as the B operand (same trick as la_out): no shuffle between the two MFMAs.
// =====================================================================================================
template <int NT>   // key tiles of 32 (N = 32 NT)
__global__ __launch_bounds__(256) void full_attn_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out) {
  constexpr int N = 32 * NT, LDK = 40, LDV = N + 8;
  __shared__ __attribute__((aligned(16))) __bf16 Ks[N * LDK];
  __shared__ __attribute__((aligned(16))) __bf16 Vt[32 * LDV];
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const bf16_t* base = qkv + (size_t)b * N * 384;
  for (int i = tid; i < N * 4; i += 256) {
    const int key = i >> 2, u = i & 3;
    *reinterpret_cast<uint4*>(Ks + key * LDK + u * 8) =
        *reinterpret_cast<const uint4*>(base + (size_t)key * 384 + 128 + h * 32 + u * 8);
    const uint4 vv = *reinterpret_cast<const uint4*>(base + (size_t)key * 384 + 256 + h * 32 + u * 8);
    const int d = key & 31;
    const int pos = (key & ~31) + (d >> 4) * 16 + ((d >> 2) & 1) * 8 + ((d >> 3) & 1) * 4 + (d & 3);
    const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      reinterpret_cast<uint16_t*>(Vt)[(u * 8 + 2 * j) * LDV + pos] = (uint16_t)(w[j] & 0xffffu);
      reinterpret_cast<uint16_t*>(Vt)[(u * 8 + 2 * j + 1) * LDV + pos] = (uint16_t)(w[j] >> 16);
    }
  }
  __syncthreads();
  for (int qt = wave; qt < NT; qt += 4) {
    const bf16_t* qp = base + (size_t)(qt * 32 + l31) * 384 + h * 32 + hi * 8;
    const bf16x8 q0 = *reinterpret_cast<const bf16x8*>(qp), q1 = *reinterpret_cast<const bf16x8*>(qp + 16);
    f32x16 sacc[NT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      sacc[kt] = zero16();
      sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(Ks + kt * 32 * LDK, LDK, l31, hi, 0), q0, sacc[kt], 0, 0, 0);
      sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(Ks + kt * 32 * LDK, LDK, l31, hi, 1), q1, sacc[kt], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sacc[kt][r] *= 0.17677669529663687f;
        m = fmaxf(m, sacc[kt][r]);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.0f;
    f32x16 oacc = zero16();
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        bf16x8 pb;
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
          const float pv = fast_exp(sacc[kt][8 * i + s2] - m);
          l += pv;
          pb[s2] = (__bf16)pv;
        }
        oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
            *reinterpret_cast<const bf16x8*>(Vt + l31 * LDV + kt * 32 + i * 16 + hi * 8), pb, oacc, 0, 0, 0);
      }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    bf16_t* op = out + ((size_t)b * N + qt * 32 + l31) * 128 + h * 32 + 4 * hi;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      uint2 w;
      w.x = pack2(oacc[4 * g4] * inv, oacc[4 * g4 + 1] * inv);
      w.y = pack2(oacc[4 * g4 + 2] * inv, oacc[4 * g4 + 3] * inv);
      *reinterpret_cast<uint2*>(op + 8 * g4) = w;
    }
  }
}


// Same contraction for N up to 1024 tokens (the 256x256 bottleneck, 32 x 32): the score tile of a (query tile, key tile)
// pair no longer stays in registers for all key tiles, so the keys are walked twice — pass 1 the row maxima (2 MFMAs per
// key tile), pass 2 the scores again, exp, row sums and V^T P (4 MFMAs) — the reference's two-pass softmax order.  K and
// V^T of one (image, head) fill the LDS (146 KB at N = 1024: one block per CU); QS blocks share a head, each takes every
// QS-th query tile so the grid covers the chip at B = 16.
template <int NT, int QS>
__global__ __launch_bounds__(256) void full_attn_mfma_big_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out) {
  constexpr int N = 32 * NT, LDK = 40, LDV = N + 8;
  extern __shared__ __attribute__((aligned(16))) char smem_attn[];
  __bf16* Ks = reinterpret_cast<__bf16*>(smem_attn);          // [N][LDK]
  __bf16* Vt = Ks + N * LDK;                                    // [32][LDV]
  const int h = blockIdx.x, b = blockIdx.y, qs = blockIdx.z, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const bf16_t* base = qkv + (size_t)b * N * 384;
  for (int i = tid; i < N * 4; i += 256) {
    const int key = i >> 2, u = i & 3;
    *reinterpret_cast<uint4*>(Ks + key * LDK + u * 8) =
        *reinterpret_cast<const uint4*>(base + (size_t)key * 384 + 128 + h * 32 + u * 8);
    const uint4 vv = *reinterpret_cast<const uint4*>(base + (size_t)key * 384 + 256 + h * 32 + u * 8);
    const int d = key & 31;
    const int pos = (key & ~31) + (d >> 4) * 16 + ((d >> 2) & 1) * 8 + ((d >> 3) & 1) * 4 + (d & 3);
    const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      reinterpret_cast<uint16_t*>(Vt)[(u * 8 + 2 * j) * LDV + pos] = (uint16_t)(w[j] & 0xffffu);
      reinterpret_cast<uint16_t*>(Vt)[(u * 8 + 2 * j + 1) * LDV + pos] = (uint16_t)(w[j] >> 16);
    }
  }
  __syncthreads();
  for (int qt = qs * 4 + wave; qt < NT; qt += 4 * QS) {
    const bf16_t* qp = base + (size_t)(qt * 32 + l31) * 384 + h * 32 + hi * 8;
    const bf16x8 q0 = *reinterpret_cast<const bf16x8*>(qp), q1 = *reinterpret_cast<const bf16x8*>(qp + 16);
    float m = -INFINITY;
#pragma unroll 2
    for (int kt = 0; kt < NT; ++kt) {
      f32x16 sa = zero16();
      sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(Ks + kt * 32 * LDK, LDK, l31, hi, 0), q0, sa, 0, 0, 0);
      sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(Ks + kt * 32 * LDK, LDK, l31, hi, 1), q1, sa, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, sa[r] * 0.17677669529663687f);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.0f;
    f32x16 oacc = zero16();
#pragma unroll 2
    for (int kt = 0; kt < NT; ++kt) {
      f32x16 sa = zero16();
      sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(Ks + kt * 32 * LDK, LDK, l31, hi, 0), q0, sa, 0, 0, 0);
      sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(Ks + kt * 32 * LDK, LDK, l31, hi, 1), q1, sa, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        bf16x8 pb;
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
          const float pv = fast_exp(sa[8 * i + s2] * 0.17677669529663687f - m);
          l += pv;
          pb[s2] = (__bf16)pv;
        }
        oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
            *reinterpret_cast<const bf16x8*>(Vt + l31 * LDV + kt * 32 + i * 16 + hi * 8), pb, oacc, 0, 0, 0);
      }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    bf16_t* op = out + ((size_t)b * N + qt * 32 + l31) * 128 + h * 32 + 4 * hi;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      uint2 w;
      w.x = pack2(oacc[4 * g4] * inv, oacc[4 * g4 + 1] * inv);
      w.y = pack2(oacc[4 * g4 + 2] * inv, oacc[4 * g4 + 3] * inv);
      *reinterpret_cast<uint2*>(op + 8 * g4) = w;
    }
  }
}

template <int C>
size_t lds_kmax() { return (size_t)kTP * Geo<C>::LDW * 2; }
template <int C>
size_t lds_out() { return (size_t)2 * kTP * Geo<C>::LDW * 2 + (size_t)kTP * kLdO * 2 + (size_t)kTP * (C / 32 > 4 ? C / 32 : 4) * 2 * 4 + (size_t)2 * C * 4; }

template <typename K>
int set_lds(K kernel, size_t bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return fail(PRG_E_HIP, std::string("hipFuncSetAttribute(fused attention): ") + hipGetErrorString(e));
  return PRG_OK;
}

int la_slabs(int N) { const int nt = ceil_div(N, kTP); return ceil_div(nt, la_tpb(nt)); }
int tail_slabs(int N) { return ceil_div(ceil_div(N, kTP), kTilesPerBlock); }

// =====================================================================================================
// ResnetBlock tail with the 1x1 res_conv folded in (sd:731-734):
//   out = SiLU(h * A[b][c] + Bc[b][c]) + Wres . cat[s0, s1] + bres          (A, Bc: GroupNorm folded by gn_coeff)
// Unfused this is a 1x1 conv launch (reads both sources, writes res) plus the flat pass (reads h and res, writes out);
// fused, res never exists in HBM: -268 MB per level-0 block.  64-pixel tiles; the source tile and the h tile go
// through LDS (coalesced 16-byte global accesses on both sides), Wres lives in registers as MFMA fragments, the
// accumulator has pixels as columns so a lane owns four consecutive channels of one pixel.
// =====================================================================================================
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void resblock_tail_fused_kernel(const bf16_t* __restrict__ h, const float* __restrict__ A,
                                                                  const float* __restrict__ Bc, const bf16_t* __restrict__ s0,
                                                                  int C0, const bf16_t* __restrict__ s1, int C1,
                                                                  const bf16_t* __restrict__ wres, const float* __restrict__ bres,
                                                                  bf16_t* __restrict__ out, int N, const float* __restrict__ head_w,
                                                                  const float* __restrict__ head_b, float* __restrict__ head_out,
                                                                  int head_sigmoid, const GnFold fold) {
  constexpr int LDX = CIN + 8, LDH = COUT + 8;
  constexpr int RT = COUT / 32, NA = RT / 2;          // row tiles of 32 channels; accumulators per wave
  constexpr int XV = CIN / 32, HV = COUT / 32;        // 16-byte vectors per thread of the source / h tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* xs = reinterpret_cast<__bf16*>(smem);       // [64][LDX]
  __bf16* hs = xs + kTP * LDX;                        // [64][LDH]
  const int b = blockIdx.y, slab = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
  const int ntiles = (N + kTP - 1) / kTP;
  const int t0 = slab * kTilesPerBlock, t1 = min(t0 + kTilesPerBlock, ntiles);
  const int yrt = RT == 2 ? (wave & 1) : wave;
  int ypt[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) ypt[a] = RT == 2 ? (wave >> 1) : a;
  bf16x8 wr[CIN / 16];
  load_wfrags<CIN>(wr, wres, yrt * 32, l31, hi);
  // this lane's 16 channels: yrt*32 + 8 g4 + 4 hi + {0..3}
  float4 ca[4], cb[4], cr[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int c0 = yrt * 32 + 8 * g4 + 4 * hi;
    if (fold.acc) {
      // GroupNorm coefficients of this lane's channel quad (inside one group) from the image's fixed-point statistics
      float mean, rstd;
      gn_fold_stats(fold, b, c0 / fold.cpg, mean, rstd);
      const float4 p4 = *reinterpret_cast<const float4*>(fold.P + (size_t)b * fold.pq_stride + c0);
      const float4 q4 = *reinterpret_cast<const float4*>(fold.Q + (size_t)b * fold.pq_stride + c0);
      ca[g4] = make_float4(rstd * p4.x, rstd * p4.y, rstd * p4.z, rstd * p4.w);
      cb[g4] = make_float4(fmaf(-mean, ca[g4].x, q4.x), fmaf(-mean, ca[g4].y, q4.y), fmaf(-mean, ca[g4].z, q4.z),
                           fmaf(-mean, ca[g4].w, q4.w));
    } else {
      ca[g4] = *reinterpret_cast<const float4*>(A + (size_t)b * COUT + c0);
      cb[g4] = *reinterpret_cast<const float4*>(Bc + (size_t)b * COUT + c0);
    }
    cr[g4] = *reinterpret_cast<const float4*>(bres + c0);
  }
  uint4 xv[XV], hv[HV];
  auto load_tile = [&](int t) {
    const int valid = min(kTP, N - t * kTP);
    const int64_t pix = (int64_t)b * N + (int64_t)t * kTP + row;
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int c = (part + 4 * i) * 8;
      xv[i] = row < valid ? (c < C0 ? *reinterpret_cast<const uint4*>(s0 + pix * C0 + c)
                                    : *reinterpret_cast<const uint4*>(s1 + pix * C1 + (c - C0)))
                          : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < HV; ++i)
      hv[i] = row < valid ? *reinterpret_cast<const uint4*>(h + pix * COUT + (part + 4 * i) * 8) : make_uint4(0, 0, 0, 0);
  };
  if (t0 < t1) load_tile(t0);
  for (int t = t0; t < t1; ++t) {
    const int valid = min(kTP, N - t * kTP);
#pragma unroll
    for (int i = 0; i < XV; ++i) *reinterpret_cast<uint4*>(xs + row * LDX + (part + 4 * i) * 8) = xv[i];
#pragma unroll
    for (int i = 0; i < HV; ++i) *reinterpret_cast<uint4*>(hs + row * LDH + (part + 4 * i) * 8) = hv[i];
    __syncthreads();                                                                            // (1) tiles staged
    if (t + 1 < t1) load_tile(t + 1);
    f32x16 ya[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) ya[a] = zero16();
#pragma unroll
    for (int kk = 0; kk < CIN / 16; ++kk)
#pragma unroll
      for (int a = 0; a < NA; ++a)
        ya[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[kk], frag(xs + ypt[a] * 32 * LDX, LDX, l31, hi, kk), ya[a], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int px = ypt[a] * 32 + l31;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        __bf16* hp = hs + px * LDH + yrt * 32 + 8 * g4 + 4 * hi;
        const uint2 hw = *reinterpret_cast<const uint2*>(hp);
        const float a4[4] = {ca[g4].x, ca[g4].y, ca[g4].z, ca[g4].w};
        const float b4[4] = {cb[g4].x, cb[g4].y, cb[g4].z, cb[g4].w};
        const float r4[4] = {cr[g4].x, cr[g4].y, cr[g4].z, cr[g4].w};
        const float hx[4] = {bf_lo(hw.x), bf_hi(hw.x), bf_lo(hw.y), bf_hi(hw.y)};
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = Elem<bf16_t>::silu(fmaf(hx[j], a4[j], b4[j])) + (ya[a][4 * g4 + j] + r4[j]);
        uint2 w;
        w.x = pack2(y[0], y[1]);
        w.y = pack2(y[2], y[3]);
        *reinterpret_cast<uint2*>(hp) = w;            // each (pixel, channel quad) is owned by exactly one lane
      }
    }
    __syncthreads();                                                                            // (2) out tile in hs
    if (head_out) {
      // the network's last layer (1x1 conv to one channel) on the tile: four lanes share a pixel's channels
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < HV; ++i) {
        const uint4 v = *reinterpret_cast<const uint4*>(hs + row * LDH + (part + 4 * i) * 8);
        const float* w = head_w + (part + 4 * i) * 8;
        const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc = fmaf(bf_lo(q[j]), w[2 * j], acc);
          acc = fmaf(bf_hi(q[j]), w[2 * j + 1], acc);
        }
      }
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      if (part == 0 && row < valid) {
        const float y = acc + head_b[0];
        head_out[(int64_t)b * N + (int64_t)t * kTP + row] = head_sigmoid ? 1.0f / (1.0f + expf(-y)) : y;
      }
    } else if (row < valid) {
#pragma unroll
      for (int i = 0; i < HV; ++i)
        *reinterpret_cast<uint4*>(out + ((int64_t)b * N + (int64_t)t * kTP + row) * COUT + (part + 4 * i) * 8) =
            *reinterpret_cast<const uint4*>(hs + row * LDH + (part + 4 * i) * 8);
    }
    __syncthreads();                                                                            // (3) tiles free
  }
}

template <int CIN, int COUT>
int launch_tail(const bf16_t* h, const float* A, const float* Bc, const bf16_t* s0, int C0, const bf16_t* s1, int C1,
                const bf16_t* wres, const float* bres, bf16_t* out, int B, int N, hipStream_t s, const GnFold& fold,
                const float* head_w = nullptr, const float* head_b = nullptr, float* head_out = nullptr, int head_sigmoid = 0) {
  const size_t lds = (size_t)kTP * (CIN + 8 + COUT + 8) * 2;
  static DeviceOnce attr;   // one-time opt-in; atomic: lanes launch from several host threads (idempotent call)
  if (!attr.done()) {
    int rc = set_lds(&resblock_tail_fused_kernel<CIN, COUT>, lds);
    if (rc) return rc;
    attr.mark();
  }
  resblock_tail_fused_kernel<CIN, COUT><<<dim3(tail_slabs(N), B), 256, lds, s>>>(h, A, Bc, s0, C0, s1, C1, wres, bres, out, N, head_w, head_b,
                                                                                 head_out, head_sigmoid, fold);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}


template <int C>
constexpr bool kPsumDefault = C >= 128;   // (measured: C = 64 58 against 70 us with it, C = 128 21.4 against 22.8, C = 256 23.7 against 25.6)

template <int C>
int launch_c(const bf16_t* x, const bf16_t* wqkv, const bf16_t* wout, const float* bias, const float* out_g, bf16_t* out,
             float* ws, int B, int N, const float* kshift, hipStream_t s) {
  static DeviceOnce attr;   // one-time opt-in; atomic: lanes launch from several host threads (idempotent call)
  if (!attr.done()) {
    int rc;
    if ((rc = set_lds(&la_kmax_fused_kernel<C>, lds_kmax<C>()))) return rc;
    if ((rc = set_lds(&la_ctx_fused_kernel<C, false, false>, lds_kmax<C>()))) return rc;
    if ((rc = set_lds(&la_ctx_fused_kernel<C, false, true>, lds_kmax<C>()))) return rc;
    if ((rc = set_lds(&la_ctx_fused_kernel<C, true, false>, lds_kmax<C>()))) return rc;
    if ((rc = set_lds(&la_ctx_fused_kernel<C, true, true>, lds_kmax<C>()))) return rc;
    if ((rc = set_lds(&la_out_fused_kernel<C, false>, lds_out<C>()))) return rc;
    if ((rc = set_lds(&la_out_fused_kernel<C, true>, lds_out<C>()))) return rc;
    attr.mark();
  }
  const int nslab = la_slabs(N);
  float* pmax = ws;
  float* ctxp = pmax + (size_t)B * nslab * kHid;
  float* sump = ctxp + (size_t)B * 4 * nslab * 1024;
  bf16_t* ctxT = reinterpret_cast<bf16_t*>(sump + (size_t)B * 4 * nslab * 32);
  const dim3 grid(nslab, B);
  if (!kshift) {
    la_kmax_fused_kernel<C><<<grid, 256, lds_kmax<C>(), s>>>(x, wqkv, pmax, N, nslab);
    PRG_LAUNCH_CHECK();
  }
  // PRG_LA_PSUM: sum_n p on the matrix pipe: -1 (default) where measured faster, 0 never, 1 always
  static const int psum_env = [] { const char* e = std::getenv("PRG_LA_PSUM"); return e ? std::atoi(e) : -1; }();
  const bool psum = psum_env > 0 || (psum_env < 0 && kPsumDefault<C>);
#define PRG_LA_CTX_GO(PS, KS) la_ctx_fused_kernel<C, PS, KS><<<grid, 256, lds_kmax<C>(), s>>>(x, wqkv, pmax, ctxp, sump, N, nslab)
  if (kshift) { if (psum) PRG_LA_CTX_GO(true, true); else PRG_LA_CTX_GO(false, true); }
  else { if (psum) PRG_LA_CTX_GO(true, false); else PRG_LA_CTX_GO(false, false); }
#undef PRG_LA_CTX_GO
  PRG_LAUNCH_CHECK();
  la_fin_fused_kernel<<<dim3(4, B), 256, 0, s>>>(ctxp, sump, ctxT, N, nslab);
  PRG_LAUNCH_CHECK();
  // (the q shifts of the four heads follow the 128 k shifts)
  const int nt = ceil_div(N, kTP);
  const dim3 ogrid(ceil_div(nt, la_out_tpb(nt)), B);
  if (kshift) la_out_fused_kernel<C, true><<<ogrid, 256, lds_out<C>(), s>>>(x, wqkv, wout, bias, out_g, ctxT, out, N, kshift + kHid);
  else la_out_fused_kernel<C, false><<<ogrid, 256, lds_out<C>(), s>>>(x, wqkv, wout, bias, out_g, ctxT, out, N, nullptr);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

}  // namespace

bool full_attention_mfma_supported(int N) { return N == 64 || N == 128 || N == 256 || N == 512 || N == 1024; }

template <int NT, int QS>
static int launch_attn_big(const bf16_t* qkv, bf16_t* out, int B, hipStream_t s) {
  constexpr int N = 32 * NT;
  constexpr size_t lds = ((size_t)N * 40 + (size_t)32 * (N + 8)) * 2;
  static DeviceOnce attr;   // one-time opt-in; atomic: lanes launch from several host threads (idempotent call)
  if (!attr.done()) {
    int rc = set_lds(&full_attn_mfma_big_kernel<NT, QS>, lds);
    if (rc) return rc;
    attr.mark();
  }
  full_attn_mfma_big_kernel<NT, QS><<<dim3(4, B, QS), 256, lds, s>>>(qkv, out);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

// qkv (B, N, 384) bf16 -> out (B, N, 128) bf16
int launch_full_attention_mfma(const bf16_t* qkv, bf16_t* out, int B, int N, hipStream_t s) {
  PRG_CHECK(full_attention_mfma_supported(N) && qkv && out, "full attention (MFMA): unsupported token count");
  const dim3 grid(4, B);
  if (N == 1024) return B >= 32 ? launch_attn_big<32, 2>(qkv, out, B, s) : launch_attn_big<32, 4>(qkv, out, B, s);
  if (N == 512) return launch_attn_big<16, 2>(qkv, out, B, s);
  if (N == 64) full_attn_mfma_kernel<2><<<grid, 256, 0, s>>>(qkv, out);
  else if (N == 128) full_attn_mfma_kernel<4><<<grid, 256, 0, s>>>(qkv, out);
  else full_attn_mfma_kernel<8><<<grid, 256, 0, s>>>(qkv, out);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

bool resblock_tail_fused_supported(int C0, int C1, int Cout) {
  const int cin = C0 + C1;
  return C0 % 8 == 0 && C1 % 8 == 0 && ((cin == 128 && Cout == 64) || (cin == 192 && Cout == 128) || (cin == 256 && Cout == 128));
}

// out (B, N, Cout) <- SiLU(h * A + Bc) + wres [Cout][C0+C1] . cat[s0 (B,N,C0), s1 (B,N,C1)] + bres.  out may alias h.
int launch_resblock_tail_fused(const bf16_t* h, const float* A, const float* Bc, const bf16_t* s0, int C0, const bf16_t* s1,
                               int C1, const bf16_t* wres, const float* bres, bf16_t* out, int B, int N, int Cout,
                               hipStream_t s,
                               const float* head_w, const float* head_b, float* head_out, int head_sigmoid, const GnFold* fold) {
  PRG_CHECK(resblock_tail_fused_supported(C0, C1, Cout) && bres, "fused resblock tail: unsupported shape");
  PRG_CHECK(!head_out || (Cout == 64 && head_w && head_b), "fused resblock tail: the head needs Cout = 64");
  GnFold f{};
  if (fold && fold->acc) {
    f = *fold;
    PRG_CHECK(f.cpg % 4 == 0 && f.G * f.cpg == Cout && f.P && f.Q, "fused resblock tail: bad GroupNorm fold");
  } else {
    PRG_CHECK(A && Bc, "fused resblock tail: no coefficients");
  }
  if (Cout == 64) return launch_tail<128, 64>(h, A, Bc, s0, C0, s1, C1, wres, bres, out, B, N, s, f, head_w, head_b, head_out, head_sigmoid);
  if (C0 + C1 == 192) return launch_tail<192, 128>(h, A, Bc, s0, C0, s1, C1, wres, bres, out, B, N, s, f);   // up level 2
  return launch_tail<256, 128>(h, A, Bc, s0, C0, s1, C1, wres, bres, out, B, N, s, f);
}

bool linattn_fused_supported(int C) { return C == 64 || C == 128 || C == 256; }

size_t linattn_fused_ws_floats(int B, int N) {
  const size_t ns = la_slabs(N);
  return (size_t)B * ns * kHid + (size_t)B * 4 * ns * 1024 + (size_t)B * 4 * ns * 32 + (size_t)B * 4 * 1024 / 2 + 64;
}

// x, out: (B, N, C) bf16.  wqkv: [384][C] with the PreNorm gain folded in (q | k | v rows, head-major); wout: [C][128].
int launch_linear_attention_fused(const bf16_t* x, const bf16_t* wqkv, const bf16_t* wout, const float* bias,
                                  const float* out_g, bf16_t* out, float* ws, int B, int N, int C, const float* kshift,
                                  hipStream_t s) {
  PRG_CHECK(linattn_fused_supported(C), "fused linear attention: unsupported width");
  PRG_CHECK(la_slabs(N) <= 4096, "fused linear attention: too many slabs");
  if (C == 64) return launch_c<64>(x, wqkv, wout, bias, out_g, out, ws, B, N, kshift, s);
  if (C == 256) return launch_c<256>(x, wqkv, wout, bias, out_g, out, ws, B, N, kshift, s);
  return launch_c<128>(x, wqkv, wout, bias, out_g, out, ws, B, N, kshift, s);
}

}  // namespace prg
