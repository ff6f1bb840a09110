// conv_epi.h — pieces shared by the convolution kernels of conv.hip and conv_split.hip: the accumulator / statistics types,
// the LDS-transposing epilogue (bias, residual, GroupNorm partial sums) and the XCD-aware tile order.
#pragma once
#include <type_traits>

#include "blocks.h"
#include "conv.h"

namespace prg {

// GroupNorm partial sums of a conv output: float in the bf16 path; the parity mode (T = float) sums in float64 — the
// variance is E[x^2] - mean^2, and fp32 sums of squares alone put ~3e-6 of error into every normalised activation
// (the reference's group_norm computes its moments to ~1e-7).
template <typename T>
using StatAcc = std::conditional_t<std::is_same<T, float>::value, double, float>;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---------------------------------------------------------------------------------------------
// shared epilogue
// ---------------------------------------------------------------------------------------------
// Wave tile = (TM*32) rows x 64 columns.  `stage` = this wave's private LDS scratch of 32 x 68 floats.
// row_to_m(local_row) -> global output row (pixel index) or -1.  Returns through (gs, gq) this lane's partial
// (sum, sumsq) over the 8 consecutive channels it stored (cols (lane & 7) * 8 .. + 7 of the wave tile).
// PART8 (f16x3 kernels): the lane's eight values of a row are summed in float32 first (16 operations), then added to the
// float64 accumulators (2 conversions + 2 additions instead of 8 + 16): a partial over 8 values carries ~1e-7 relative error,
// which the sum over a group's >= 2048 partials averages out far below the 1e-7 the moments need.
template <typename T, int TM, typename RowMap, bool PART8 = false>
__device__ inline void epilogue_store(const ConvLaunch<T>& L, f32x16 (&acc)[TM][2], float* stage, int lane,
                                      int col0, RowMap row_to_m, StatAcc<T>& gs, StatAcc<T>& gq) {
  using SA = StatAcc<T>;
  constexpr int P = 68;  // stage pitch (floats): 64 + 4 keeps rows 16-byte aligned and off one bank
  constexpr int VEC = Elem<T>::kVec;
  const int l31 = lane & 31, hi = lane >> 5;
  const int cc = lane & 7, rr = lane >> 3;  // read-back role: 8 lanes per row, 8 rows per pass
  const int col = col0 + cc * 8;
  const bool col_ok = col < L.d.Cout;       // Cout is a multiple of 8 on this path
  float bias[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) bias[u] = (L.bias && col_ok) ? L.bias[col + u] : 0.0f;
  gs = 0;
  gq = 0;
  // activated residual with in-kernel GroupNorm fold (common.h, GnFold): this thread's 8 channels lie in one group; the
  // coefficients are recomputed only when the row's image changes (a tile spans one image, rarely two)
  float fa[8], fb[8];
  int fimg = -1;
  // Round 5: the row's image index m / (Hout * Wout) was a 32-bit integer DIVISION per row and lane (~25 VALU instructions, seven of
  // them quarter-rate multiplies) — with a power-of-two image (every shape the networks produce) it is a shift; and the coefficient
  // TABLES of the activated residual (res_a / res_b: 64 bytes per row and lane) are re-read only when the row's image changes, as the
  // in-kernel fold already did.  The 1x1 kernels' epilogue — not their operand fetch — was what kept them at 0.18 MFMA-busy.
  const int hw_ = L.d.Hout * L.d.Wout;
  const int hw_sh = (hw_ & (hw_ - 1)) == 0 ? 31 - __builtin_clz((unsigned)hw_) : -1;
  auto img_of = [&](int64_t m) -> int { return hw_sh >= 0 ? (int)((unsigned)m >> hw_sh) : (int)m / hw_; };
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    __syncthreads();  // previous pass fully read (and, first time, the main loop's LDS reads are done)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) stage[((e & 3) + 8 * (e >> 2) + 4 * hi) * P + j * 32 + l31] = acc[i][j][e];
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = pass * 8 + rr;
      const int64_t m = row_to_m(i * 32 + r);
      const float4 v0 = *reinterpret_cast<const float4*>(stage + r * P + cc * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(stage + r * P + cc * 8 + 4);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      if (m >= 0 && col_ok) {
        const size_t o = (size_t)m * L.d.Cout + col;
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] += bias[u];
        if constexpr (PART8) {
          float s8 = 0.0f, q8 = 0.0f;
#pragma unroll
          for (int u = 0; u < 8; ++u) { s8 += v[u]; q8 = fmaf(v[u], v[u], q8); }
          gs += (SA)s8;
          gq += (SA)q8;
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if constexpr (std::is_same<SA, double>::value) { gs += (double)v[u]; gq += (double)v[u] * (double)v[u]; }
            else { gs += v[u]; gq = fmaf(v[u], v[u], gq); }
          }
        }
        if (L.residual) {
          float ra[8], rb[8];
          const bool act = L.res_a != nullptr || L.res_fold.acc != nullptr;
          if (L.res_fold.acc) {
            const int img = img_of(m);
            if (img != fimg) {
              fimg = img;
              const GnFold& f = L.res_fold;
              float mean, rstd;
              gn_fold_stats(f, img, col / f.cpg, mean, rstd);
              const float* pp = f.P + (size_t)img * f.pq_stride + col;
              const float* pq = f.Q + (size_t)img * f.pq_stride + col;
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                fa[u] = rstd * pp[u];
                fb[u] = fmaf(-mean, fa[u], pq[u]);
              }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { ra[u] = fa[u]; rb[u] = fb[u]; }
          } else if (L.res_a) {
            const int img = img_of(m);                                                     // this row's image
            if (img != fimg) {
              fimg = img;
              const size_t cb = (size_t)img * L.d.Cout + col;
              const float4 a0 = *reinterpret_cast<const float4*>(L.res_a + cb), a1 = *reinterpret_cast<const float4*>(L.res_a + cb + 4);
              const float4 b0 = *reinterpret_cast<const float4*>(L.res_b + cb), b1 = *reinterpret_cast<const float4*>(L.res_b + cb + 4);
              fa[0] = a0.x; fa[1] = a0.y; fa[2] = a0.z; fa[3] = a0.w; fa[4] = a1.x; fa[5] = a1.y; fa[6] = a1.z; fa[7] = a1.w;
              fb[0] = b0.x; fb[1] = b0.y; fb[2] = b0.z; fb[3] = b0.w; fb[4] = b1.x; fb[5] = b1.y; fb[6] = b1.z; fb[7] = b1.w;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { ra[u] = fa[u]; rb[u] = fb[u]; }
          }
#pragma unroll
          for (int h = 0; h < 8 / VEC; ++h) {
            Vec16<T> rv = vec_load(L.residual + o + h * VEC);
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
              const float r = Elem<T>::load(rv.e[u]);
              v[h * VEC + u] += act ? Elem<T>::silu(fmaf(r, ra[h * VEC + u], rb[h * VEC + u])) : r;
            }
          }
        }
#pragma unroll
        for (int h = 0; h < 8 / VEC; ++h) {
          Vec16<T> w;
#pragma unroll
          for (int u = 0; u < VEC; ++u) w.e[u] = Elem<T>::store(v[h * VEC + u]);
          vec_store(L.out + o + h * VEC, w);
        }
      }
    }
  }
}

// Block-level, fixed-order reduction of the lanes' (gs, gq) into per-group partials and one global store per group.
// red = LDS scratch of 4 waves x 8 column chunks x 2 floats.  Wave layout WAVES_M x WAVES_N, wave tile 64 columns.
template <int WAVES_M, int WAVES_N, typename SA>
__device__ inline void epilogue_stats(SA* red, SA gs, SA gq, int wave, int lane, int tn_col0, int Cout,
                                      int groups, float* dst /* partials + (image*nsplit + slab) * groups * 2 */) {
  // lanes with equal (lane & 7) stored the same 8-channel column chunk: fold the 8 row-lanes together
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    gs += __shfl_xor(gs, o, 64);
    gq += __shfl_xor(gq, o, 64);
  }
  __syncthreads();
  if (lane < 8) {
    red[(wave * 8 + lane) * 2 + 0] = gs;
    red[(wave * 8 + lane) * 2 + 1] = gq;
  }
  __syncthreads();
  const int cpg = Cout / groups;                 // multiple of 8 on this path
  const int ngrp_blk = (WAVES_N * 64) / cpg;     // groups covered by this workgroup's columns
  const int t = wave * 64 + lane;
  if (t < ngrp_blk) {
    const int g = tn_col0 / cpg + t;
    if (g < groups) {
      SA ss = 0, qq = 0;
      for (int ch = 0; ch < cpg / 8; ++ch) {
        const int cblk = t * (cpg / 8) + ch;     // 8-channel chunk index within the workgroup's columns
        const int wn = cblk / 8, c8 = cblk % 8;
        for (int wm = 0; wm < WAVES_M; ++wm) {
          ss += red[((wm * WAVES_N + wn) * 8 + c8) * 2 + 0];
          qq += red[((wm * WAVES_N + wn) * 8 + c8) * 2 + 1];
        }
      }
      dst[g * 2 + 0] = (float)ss;
      dst[g * 2 + 1] = (float)qq;
    }
  }
}

// XCD-aware tile order: workgroup b runs on XCD b % 8 (observed); give each XCD a contiguous run of tiles so the
// tiles that share input halos / weight tiles meet in one L2.  Bijective for any grid size.
__device__ inline int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// LDS the epilogue needs: four waves' 32 x 68 float stages + the statistics scratch
constexpr size_t kEpilogueLds = (size_t)4 * 32 * 68 * sizeof(float) + 4 * 8 * 2 * sizeof(double);  // 35,328 B

}  // namespace prg
