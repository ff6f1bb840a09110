// blocks.hip — the HBM-bound kernels between the convolutions (NHWC activations of type T, fp32 math):
// GroupNorm statistics + apply (+conditioning, SiLU, residual), channel LayerNorm (+residual), the linear-attention
// core, the bottleneck attention core, and the tiny conditioning MLPs.  Reductions are wave-level (64 lanes) and
// every cross-workgroup reduction is done in a fixed order, so results are run-to-run deterministic.
//
// sd = denoising_diffusion_pytorch/successive_ddnm_diffusion.py
#include <atomic>

#include "blocks.h"

namespace prg {

template <typename T>
struct Acc {  // statistics accumulator: float32 activations (parity mode) accumulate in float64
  using type = float;
};
template <>
struct Acc<float> {
  using type = double;
};

static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// =============================================================================================
// GroupNorm
// =============================================================================================
// grid (nsplit, B); thread owns channel-vector `vc = tid % (C/VEC)` and walks pixels with stride 256/(C/VEC)
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, float* __restrict__ partials, int HW,
                                                       int C, int G, int slab) {
  constexpr int VEC = Elem<T>::kVec;
  using A = typename Acc<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  A* red = reinterpret_cast<A*>(smem);  // [psub][C][2]
  const int b = blockIdx.y, sp = blockIdx.x, nsplit = gridDim.x;
  const int nvc = C / VEC, psub = 256 / nvc;
  const int vc = threadIdx.x % nvc, ps = threadIdx.x / nvc;
  const int p0 = sp * slab, p1 = min(HW, p0 + slab);
  A s[VEC], q[VEC];
#pragma unroll
  for (int u = 0; u < VEC; ++u) { s[u] = 0; q[u] = 0; }
  const T* xb = x + (size_t)b * HW * C;
  for (int p = p0 + ps; p < p1; p += psub) {
    Vec16<T> v = vec_load(xb + (size_t)p * C + vc * VEC);
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      A f = (A)Elem<T>::load(v.e[u]);
      s[u] += f;
      q[u] += f * f;
    }
  }
#pragma unroll
  for (int u = 0; u < VEC; ++u) {
    red[((size_t)ps * C + vc * VEC + u) * 2 + 0] = s[u];
    red[((size_t)ps * C + vc * VEC + u) * 2 + 1] = q[u];
  }
  __syncthreads();
  // fixed-order reduction: channel c over pixel sub-lanes, then group g over its channels
  for (int c = threadIdx.x; c < C; c += 256) {
    A ss = 0, qq = 0;
    for (int k = 0; k < psub; ++k) {
      ss += red[((size_t)k * C + c) * 2 + 0];
      qq += red[((size_t)k * C + c) * 2 + 1];
    }
    red[(size_t)c * 2 + 0] = ss;  // row 0 of red reused (each thread only overwrites the column it just summed)
    red[(size_t)c * 2 + 1] = qq;
  }
  __syncthreads();
  if (threadIdx.x < G) {
    const int cpg = C / G, g = threadIdx.x;
    A ss = 0, qq = 0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      ss += red[(size_t)c * 2 + 0];
      qq += red[(size_t)c * 2 + 1];
    }
    float* o = partials + (((size_t)b * nsplit + sp) * G + g) * 2;
    o[0] = (float)ss;
    o[1] = (float)qq;
  }
}

template <typename T>
int launch_gn_stats(const T* x, float* partials, int B, int HW, int C, int G, int* nsplit, hipStream_t s) {
  constexpr int VEC = Elem<T>::kVec;
  using A = typename Acc<T>::type;
  const int nvc = C / VEC;
  PRG_CHECK(C % VEC == 0 && is_pow2(nvc) && nvc <= 256, "groupnorm: C/VEC must be a power of two <= 256");
  PRG_CHECK(C % G == 0 && G <= 64, "groupnorm: bad group count");
  const int psub = 256 / nvc;
  // slabs of >= 8 pixel rows per thread, at most kGnMaxSplit slabs per image
  int ns = ceil_div(HW, psub * 8);
  if (ns > kGnMaxSplit) ns = kGnMaxSplit;
  if (ns < 1) ns = 1;
  const int slab = ceil_div(HW, ns);
  ns = ceil_div(HW, slab);
  *nsplit = ns;
  const size_t lds = (size_t)psub * C * 2 * sizeof(A);
  gn_stats_kernel<T><<<dim3(ns, B), 256, lds, s>>>(x, partials, HW, C, G, slab);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}
template int launch_gn_stats<float>(const float*, float*, int, int, int, int, int*, hipStream_t);
template int launch_gn_stats<bf16_t>(const bf16_t*, float*, int, int, int, int, int*, hipStream_t);

// grid (chunks, B).  y = silu( gn(x) * (scale + 1) + shift ) + residual
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ partials,
                                                       int nsplit, GnApply p, const T* __restrict__ residual,
                                                       T* __restrict__ out, int HW, int C, int G, int slab) {
  constexpr int VEC = Elem<T>::kVec;
  __shared__ float s_mean[64], s_rstd[64];
  const int b = blockIdx.y;
  if (threadIdx.x < G) {
    double ss = 0, qq = 0;
    const float* pp = partials + ((size_t)b * nsplit * G + threadIdx.x) * 2;
    for (int k = 0; k < nsplit; ++k) {
      ss += (double)pp[(size_t)k * G * 2 + 0];
      qq += (double)pp[(size_t)k * G * 2 + 1];
    }
    const double n = (double)HW * (C / G);
    const double mean = ss / n;
    double var = qq / n - mean * mean;
    if (var < 0) var = 0;
    s_mean[threadIdx.x] = (float)mean;
    s_rstd[threadIdx.x] = (float)(1.0 / sqrt(var + 1e-5));
  }
  __syncthreads();
  const int nvc = C / VEC, psub = 256 / nvc;
  const int vc = threadIdx.x % nvc, ps = threadIdx.x / nvc;
  const int cpg = C / G;
  float a[VEC], bb[VEC], sc[VEC], sh[VEC];
  const float* ssa = nullptr;
  const float* ssb = nullptr;
  if (p.ss_a) {
    ssa = p.ss_a + (size_t)b * p.ss_a_stride;
    if (p.ss_a_row) ssa += (size_t)(*p.ss_a_row) * p.ss_a_row_stride;
    if (p.ss_b) ssb = p.ss_b + (size_t)b * p.ss_b_stride;
  }
#pragma unroll
  for (int u = 0; u < VEC; ++u) {
    const int c = vc * VEC + u, g = c / cpg;
    const float scale = s_rstd[g] * p.gamma[c];
    a[u] = scale;
    bb[u] = p.beta[c] - s_mean[g] * scale;
    sc[u] = 1.0f;
    sh[u] = 0.0f;
    if (ssa) {
      float s0 = ssa[c], s1 = ssa[C + c];
      if (ssb) { s0 += ssb[c]; s1 += ssb[C + c]; }
      sc[u] = s0 + 1.0f;
      sh[u] = s1;
    }
  }
  const int p0 = blockIdx.x * slab, p1 = min(HW, p0 + slab);
  const size_t base = (size_t)b * HW * C;
#pragma unroll 4
  for (int px = p0 + ps; px < p1; px += psub) {
    const size_t o = base + (size_t)px * C + vc * VEC;
    Vec16<T> v = vec_load(x + o), r, w;
    if (residual) r = vec_load(residual + o);
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      float y = fmaf(Elem<T>::load(v.e[u]), a[u], bb[u]);
      if (ssa) y = fmaf(y, sc[u], sh[u]);
      y = Elem<T>::silu(y);
      if (residual) y += Elem<T>::load(r.e[u]);
      w.e[u] = Elem<T>::store(y);
    }
    vec_store(out + o, w);
  }
}

template <typename T>
int launch_gn_apply(const T* x, const float* partials, int nsplit, const GnApply& p, const T* residual, T* out, int B,
                    int HW, int C, int G, hipStream_t s) {
  constexpr int VEC = Elem<T>::kVec;
  const int nvc = C / VEC;
  PRG_CHECK(C % VEC == 0 && is_pow2(nvc) && nvc <= 256, "groupnorm: C/VEC must be a power of two <= 256");
  const int psub = 256 / nvc;
  int chunks = ceil_div(HW, psub * 4);
  if (chunks > 512) chunks = 512;
  if (chunks < 1) chunks = 1;
  const int slab = ceil_div(HW, chunks);
  chunks = ceil_div(HW, slab);
  gn_apply_kernel<T><<<dim3(chunks, B), 256, 0, s>>>(x, partials, nsplit, p, residual, out, HW, C, G, slab);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}
template int launch_gn_apply<float>(const float*, const float*, int, const GnApply&, const float*, float*, int, int, int,
                                    int, hipStream_t);
template int launch_gn_apply<bf16_t>(const bf16_t*, const float*, int, const GnApply&, const bf16_t*, bf16_t*, int, int,
                                     int, int, hipStream_t);

// Flat elementwise pass: y = silu(x * A[b][c] + Bc[b][c]) + residual, with the coefficients precomputed by
// gn_coeff_kernel (no per-workgroup statistics prologue: every workgroup streams from its first instruction).
template <typename T>
__global__ __launch_bounds__(256) void affine_silu_kernel(const T* __restrict__ x, const float* __restrict__ A,
                                                          const float* __restrict__ Bc, const T* __restrict__ residual,
                                                          T* __restrict__ out, int64_t nvec_per_img, int nvc, int C,
                                                          int64_t total) {
  constexpr int VEC = Elem<T>::kVec;
  // consecutive threads -> consecutive 16-byte vectors; a thread's channel vector index is fixed when the grid stride
  // is a multiple of nvc (it is: 256 * gridDim.x with nvc a power of two <= 256)
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += stride * 4) {
    Vec16<T> v[4], r[4];
    int64_t idx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      idx[k] = i0 + k * stride;
      if (idx[k] < total) {
        v[k] = vec_load(x + idx[k] * VEC);
        if (residual) r[k] = vec_load(residual + idx[k] * VEC);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (idx[k] < total) {
        const int b = (int)(idx[k] / nvec_per_img);
        const int vc = (int)(idx[k] % nvc);
        const float* a = A + (size_t)b * C + vc * VEC;
        const float* bb = Bc + (size_t)b * C + vc * VEC;
        Vec16<T> w;
#pragma unroll
        for (int u = 0; u < VEC; ++u) {
          float y = Elem<T>::silu(fmaf(Elem<T>::load(v[k].e[u]), a[u], bb[u]));
          if (residual) y += Elem<T>::load(r[k].e[u]);
          w.e[u] = Elem<T>::store(y);
        }
        vec_store(out + idx[k] * VEC, w);
      }
    }
  }
}

template <typename T>
int launch_affine_silu(const T* x, const float* A, const float* Bc, const T* residual, T* out, int B, int HW, int C,
                       hipStream_t s) {
  constexpr int VEC = Elem<T>::kVec;
  PRG_CHECK(C % VEC == 0, "affine_silu: C must be a multiple of the vector width");
  const int nvc = C / VEC;
  const int64_t per_img = (int64_t)HW * nvc, total = per_img * B;
  int64_t blocks = (total + 1023) / 1024;
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  affine_silu_kernel<T><<<(int)blocks, 256, 0, s>>>(x, A, Bc, residual, out, per_img, nvc, C, total);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}
template int launch_affine_silu<float>(const float*, const float*, const float*, const float*, float*, int, int, int,
                                       hipStream_t);
template int launch_affine_silu<bf16_t>(const bf16_t*, const float*, const float*, const bf16_t*, bf16_t*, int, int, int,
                                        hipStream_t);

// grid (B): GroupNorm (+ conditioning) folded to y = x * A[b][c] + Bc[b][c].  38 launches per U-Net evaluation, each a
// short dependent chain: the per-channel parameters (norm gain / bias, the conditioning row found through the device step
// counter) are fetched BEFORE the statistics so their latency runs under the partial-sum loads.
__global__ __launch_bounds__(256) void gn_coeff_kernel(const float* __restrict__ partials, int nsplit, GnApply p,
                                                       float* __restrict__ A, float* __restrict__ Bc, int HW, int C,
                                                       int G) {
  __shared__ float s_mean[64], s_rstd[64];
  const int b = blockIdx.x;
  // per-channel parameters of up to four channels per thread (C <= 1024)
  const float* ssa = nullptr;
  const float* ssb = nullptr;
  if (p.ss_a) {
    ssa = p.ss_a + (size_t)b * p.ss_a_stride;
    if (p.ss_a_row) ssa += (size_t)(*p.ss_a_row) * p.ss_a_row_stride;
    if (p.ss_b) ssb = p.ss_b + (size_t)b * p.ss_b_stride;
  }
  float gam[4] = {0, 0, 0, 0}, bet[4] = {0, 0, 0, 0}, s0v[4] = {0, 0, 0, 0}, s1v[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = threadIdx.x + 256 * k;
    if (c < C) {
      gam[k] = p.gamma[c];
      bet[k] = p.beta[c];
      if (ssa) {
        s0v[k] = ssa[c];
        s1v[k] = ssa[C + c];
        if (ssb) { s0v[k] += ssb[c]; s1v[k] += ssb[C + c]; }
      }
    }
  }
  // LPG lanes (a power of two <= 64) share one group: strided partial sums, then a fixed shuffle tree (deterministic)
  int LPG = 64;
  while (LPG * G > 256) LPG >>= 1;
  const int g = threadIdx.x / LPG, j = threadIdx.x % LPG;
  if (g < G) {
    double ss = 0, qq = 0;
    const float* pp = partials + ((size_t)b * nsplit * G + g) * 2;
    for (int k0 = j; k0 < nsplit; k0 += 8 * LPG) {   // eight loads in flight per lane, summed in index order
      float2 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = k0 + i * LPG;
        v[i] = k < nsplit ? *reinterpret_cast<const float2*>(pp + (size_t)k * G * 2) : make_float2(0.0f, 0.0f);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ss += (double)v[i].x;
        qq += (double)v[i].y;
      }
    }
    for (int o = LPG >> 1; o > 0; o >>= 1) {
      ss += __shfl_xor(ss, o, 64);
      qq += __shfl_xor(qq, o, 64);
    }
    if (j == 0) {
      const double n = (double)HW * (C / G);
      const double mean = ss / n;
      double var = qq / n - mean * mean;
      if (var < 0) var = 0;
      s_mean[g] = (float)mean;
      s_rstd[g] = (float)(1.0 / sqrt(var + 1e-5));
    }
  }
  __syncthreads();
  const int cpg = C / G;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = threadIdx.x + 256 * k;
    if (c < C) {
      const int gg = c / cpg;
      float a = s_rstd[gg] * gam[k];
      float bb = bet[k] - s_mean[gg] * a;
      if (ssa) {
        a = a * (s0v[k] + 1.0f);
        bb = fmaf(bb, s0v[k] + 1.0f, s1v[k]);
      }
      A[(size_t)b * C + c] = a;
      Bc[(size_t)b * C + c] = bb;
    }
  }
}

int launch_gn_coeff(const float* partials, int nsplit, const GnApply& p, float* A, float* Bc, int B, int HW, int C,
                    int G, hipStream_t s) {
  PRG_CHECK(partials && A && Bc && G <= 64 && C % G == 0 && C <= 1024, "gn_coeff: bad arguments");
  gn_coeff_kernel<<<B, 256, 0, s>>>(partials, nsplit, p, A, Bc, HW, C, G);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

// ---------------------------------------------------------------------------------------------
// Fixed-point GroupNorm statistics (common.h, GnFold): the three small kernels around the in-kernel folds
// ---------------------------------------------------------------------------------------------
// grid (B): coefficient tables from the accumulators, for the consumers that cannot fold in-kernel
__global__ __launch_bounds__(256) void gn_coeff_acc_kernel(GnFold f, float* __restrict__ A, float* __restrict__ Bc, int C) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    float mean, rstd;
    gn_fold_stats(f, b, c / f.cpg, mean, rstd);
    const float a = rstd * f.P[(size_t)b * f.pq_stride + c];
    A[(size_t)b * C + c] = a;
    Bc[(size_t)b * C + c] = fmaf(-mean, a, f.Q[(size_t)b * f.pq_stride + c]);
  }
}

int launch_gn_coeff_acc(const GnFold& f, float* A, float* Bc, int B, int C, hipStream_t s) {
  PRG_CHECK(f.acc && f.P && f.Q && A && Bc && f.G * f.cpg == C, "gn_coeff_acc: bad arguments");
  gn_coeff_acc_kernel<<<B, 256, 0, s>>>(f, A, Bc, C);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

// grid (entries, B): P = gamma (scale + 1), Q = beta (scale + 1) + shift of every conditioned GroupNorm of one forward
// `zero` (may be null): the evaluation's fixed-point GroupNorm accumulators, cleared here instead of by a separate memset launch
// (round 6: one launch fewer per evaluation; every later kernel of the forward is behind this one on the stream)
__global__ __launch_bounds__(256) void cond_fold_kernel(const CondFoldEntry* __restrict__ entries, const float* __restrict__ flat,
                                                        GnApply ss, float* __restrict__ pq, int64_t pq_stride,
                                                        long long* __restrict__ zero, int64_t zero_words) {
  const CondFoldEntry e = entries[blockIdx.x];
  const int b = blockIdx.y;
  if (zero) {
    const int64_t nthr = (int64_t)gridDim.x * gridDim.y * 256;
    for (int64_t i = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < zero_words; i += nthr) zero[i] = 0;
  }
  const float* ssa = ss.ss_a + (size_t)b * ss.ss_a_stride + e.ss_off;
  if (ss.ss_a_row) ssa += (size_t)(*ss.ss_a_row) * ss.ss_a_row_stride;
  const float* ssb = ss.ss_b ? ss.ss_b + (size_t)b * ss.ss_b_stride + e.ss_off : nullptr;
  const float* gamma = flat + e.gamma_off;
  const float* beta = flat + e.beta_off;
  float* o = pq + (size_t)b * pq_stride + e.ss_off;
  for (int c = threadIdx.x; c < e.C; c += 256) {
    float s0 = ssa[c], s1 = ssa[e.C + c];
    if (ssb) { s0 += ssb[c]; s1 += ssb[e.C + c]; }
    o[c] = gamma[c] * (s0 + 1.0f);
    o[e.C + c] = fmaf(beta[c], s0 + 1.0f, s1);
  }
}

int launch_cond_fold(const CondFoldEntry* entries, int n, const float* flat, const GnApply& ss, float* pq, int64_t pq_stride,
                     int B, hipStream_t s, long long* zero, int64_t zero_words) {
  PRG_CHECK(entries && n > 0 && flat && ss.ss_a && pq, "cond_fold: bad arguments");
  cond_fold_kernel<<<dim3(n, B), 256, 0, s>>>(entries, flat, ss, pq, pq_stride, zero, zero_words);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

// grid (blocks per image, B): y = silu(x * A + B) + residual; every block folds its image's coefficients into LDS first
// (C <= 1024: two int64 loads and two parameter loads per channel), then streams a contiguous run of the image's vectors
__global__ __launch_bounds__(256) void affine_silu_fold_kernel(const bf16_t* __restrict__ x, GnFold f,
                                                               const bf16_t* __restrict__ residual, bf16_t* __restrict__ out,
                                                               int64_t nvec_per_img, int nvc, int C) {
  __shared__ float sA[1024], sB[1024];
  const int b = blockIdx.y;
  for (int c = threadIdx.x; c < C; c += 256) {
    float mean, rstd;
    gn_fold_stats(f, b, c / f.cpg, mean, rstd);
    const float a = rstd * f.P[(size_t)b * f.pq_stride + c];
    sA[c] = a;
    sB[c] = fmaf(-mean, a, f.Q[(size_t)b * f.pq_stride + c]);
  }
  __syncthreads();
  const int64_t per_blk = (nvec_per_img + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per_blk, hi = min(lo + per_blk, nvec_per_img);
  const int64_t base = (int64_t)b * nvec_per_img;
  for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += 1024) {
    Vec16<bf16_t> v[4], r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = i0 + k * 256;
      if (i < hi) {
        v[k] = vec_load(x + (base + i) * 8);
        if (residual) r[k] = vec_load(residual + (base + i) * 8);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = i0 + k * 256;
      if (i < hi) {
        const int vc = (int)(i % nvc);
        Vec16<bf16_t> w;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          float y = Elem<bf16_t>::silu(fmaf(Elem<bf16_t>::load(v[k].e[u]), sA[vc * 8 + u], sB[vc * 8 + u]));
          if (residual) y += Elem<bf16_t>::load(r[k].e[u]);
          w.e[u] = Elem<bf16_t>::store(y);
        }
        vec_store(out + (base + i) * 8, w);
      }
    }
  }
}

int launch_affine_silu_fold(const bf16_t* x, const GnFold& f, const bf16_t* residual, bf16_t* out, int B, int HW, int C,
                            hipStream_t s) {
  PRG_CHECK(f.acc && f.P && f.Q && C % 8 == 0 && C <= 1024 && f.G * f.cpg == C, "affine_silu_fold: bad arguments");
  const int nvc = C / 8;
  const int64_t per_img = (int64_t)HW * nvc;
  // ~4096 vectors (64 KB in, 64 KB out) per block, at least one block per image, at most 16384 blocks in all; small tensors (the
  // 32x32 and 16x16 levels at B = 64: 128-256 blocks of four serial iterations, 13 us for 25-50 MB) take 1024-vector blocks: one
  // iteration per thread, 512-1024 blocks
  const int64_t vpb = per_img * B < ((int64_t)2 << 20) ? 1024 : 4096;
  int64_t bpi = (per_img + vpb - 1) / vpb;
  if (bpi * B > 16384) bpi = 16384 / B > 0 ? 16384 / B : 1;
  if (bpi < 1) bpi = 1;
  affine_silu_fold_kernel<<<dim3((int)bpi, B), 256, 0, s>>>(x, f, residual, out, per_img, nvc, C);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

// =============================================================================================
// channel LayerNorm (per pixel over C, biased variance, eps 1e-5, gain only) + optional residual
// =============================================================================================
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const float* __restrict__ g,
                                                        const T* __restrict__ residual, T* __restrict__ out,
                                                        int64_t M, int C, int L) {
  constexpr int VEC = Elem<T>::kVec;
  // L lanes (power of two <= 64) share one pixel; each holds up to MAXV 16-byte vectors (stride L)
  const int sub = threadIdx.x % L;
  const int per_block = 256 / L;
  const int nv = C / VEC;
  for (int64_t m = (int64_t)blockIdx.x * per_block + threadIdx.x / L; m < M; m += (int64_t)gridDim.x * per_block) {
    float v[MAXV][VEC];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int iv = sub + k * L;
      if (iv < nv) {
        Vec16<T> t = vec_load(x + m * C + (size_t)iv * VEC);
#pragma unroll
        for (int u = 0; u < VEC; ++u) { v[k][u] = Elem<T>::load(t.e[u]); s += v[k][u]; }
      }
    }
    for (int o = L >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int iv = sub + k * L;
      if (iv < nv) {
#pragma unroll
        for (int u = 0; u < VEC; ++u) { float dlt = v[k][u] - mean; q = fmaf(dlt, dlt, q); }
      }
    }
    for (int o = L >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / (float)C + 1e-5f);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int iv = sub + k * L;
      if (iv < nv) {
        const size_t o = (size_t)m * C + (size_t)iv * VEC;
        Vec16<T> w, r;
        if (residual) r = vec_load(residual + o);
#pragma unroll
        for (int u = 0; u < VEC; ++u) {
          float y = (v[k][u] - mean) * rstd * g[iv * VEC + u];
          if (residual) y += Elem<T>::load(r.e[u]);
          w.e[u] = Elem<T>::store(y);
        }
        vec_store(out + o, w);
      }
    }
  }
}

template <typename T>
int launch_layernorm(const T* x, const float* g, const T* residual, T* out, int64_t M, int C, hipStream_t s) {
  constexpr int VEC = Elem<T>::kVec;
  PRG_CHECK(C % VEC == 0, "layernorm: C must be a multiple of the vector width");
  const int nv = C / VEC;
  int L = 1;
  while (L < nv && L < 64) L <<= 1;
  const int need = ceil_div(nv, L);
  PRG_CHECK(need <= 4, "layernorm: C too large");
  const int per_block = 256 / L;
  int grid = (int)((M + per_block - 1) / per_block);
  if (grid > 16384) grid = 16384;
  if (need <= 1)
    layernorm_kernel<T, 1><<<grid, 256, 0, s>>>(x, g, residual, out, M, C, L);
  else if (need <= 2)
    layernorm_kernel<T, 2><<<grid, 256, 0, s>>>(x, g, residual, out, M, C, L);
  else
    layernorm_kernel<T, 4><<<grid, 256, 0, s>>>(x, g, residual, out, M, C, L);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}
template int launch_layernorm<float>(const float*, const float*, const float*, float*, int64_t, int, hipStream_t);
template int launch_layernorm<bf16_t>(const bf16_t*, const float*, const bf16_t*, bf16_t*, int64_t, int, hipStream_t);

// =============================================================================================
// linear attention core (sd:755-768).  qkv (B, N, 384): q = [0,128) k = [128,256) v = [256,384), head-major.
//   k' = softmax_n(k)   q' = softmax_d(q) / sqrt(32)   v' = v / N
//   ctx[d][e] = sum_n k'[n][d] v'[n][e]     out[n][e] = sum_d ctx[d][e] q'[n][d]
// ws layout (floats): kmax [B][128] | ctxp [B][4][NS][1024] | sump [B][4][NS][32] | ctx [B][4][1024]
// =============================================================================================
constexpr int kLaMaxSplit = 128;
static inline int la_nsplit(int N) {
  int ns = ceil_div(N, 512);    // >= 512 pixels per slab: the fixed-order reduce of the slabs stays cheap
  if (ns > kLaMaxSplit) ns = kLaMaxSplit;
  return ns < 1 ? 1 : ns;
}
size_t linattn_ws_floats(int B, int N) {
  const size_t ns = la_nsplit(N);
  return (size_t)B * 128 + (size_t)B * 4 * ns * 1024 + (size_t)B * 4 * ns * 32 + (size_t)B * 4 * 1024;
}

__device__ inline void atomic_max_f32(float* addr, float v) {
  if (v >= 0.0f)
    atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else
    atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ void fill_u32_kernel(uint32_t* p, uint32_t v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// grid (slabs, B): column max of k.  16-byte loads: thread = (pixel lane 0..15, 8-channel vector 0..15), running max
// over the slab in registers, LDS fold over the 16 pixel lanes, then ONE order-independent atomic max per channel.
template <typename T>
__global__ __launch_bounds__(256) void la_kmax_kernel(const T* __restrict__ qkv, float* __restrict__ kmax, int N,
                                                      int slab) {
  constexpr int VEC = Elem<T>::kVec;
  constexpr int NV = 128 / VEC;              // vectors per pixel (bf16 16, f32 32)
  constexpr int PL = 256 / NV;               // pixel lanes (16 / 8)
  __shared__ float sm[PL][128];
  const int b = blockIdx.y, vc = threadIdx.x % NV, pl = threadIdx.x / NV;
  const int p0 = blockIdx.x * slab, p1 = min(N, p0 + slab);
  float m[VEC];
#pragma unroll
  for (int u = 0; u < VEC; ++u) m[u] = -INFINITY;
  const T* base = qkv + (size_t)b * N * 384 + 128 + vc * VEC;
#pragma unroll 4
  for (int n = p0 + pl; n < p1; n += PL) {
    Vec16<T> v = vec_load(base + (size_t)n * 384);
#pragma unroll
    for (int u = 0; u < VEC; ++u) m[u] = fmaxf(m[u], Elem<T>::load(v.e[u]));
  }
#pragma unroll
  for (int u = 0; u < VEC; ++u) sm[pl][vc * VEC + u] = m[u];
  __syncthreads();
  if (threadIdx.x < 128) {
    float r = sm[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < PL; ++k) r = fmaxf(r, sm[k][threadIdx.x]);
    if (r > -INFINITY) atomic_max_f32(kmax + (size_t)b * 128 + threadIdx.x, r);
  }
}

// grid (ns, 4, B): partial context of one head over a slab: thread (d = tid/8, e0 = (tid%8)*4).
// Staging with 16-byte loads: thread = (pixel 0..P-1, vector of the head's 32 channels).
template <typename T>
__global__ __launch_bounds__(256) void la_ctx_kernel(const T* __restrict__ qkv, const float* __restrict__ kmax,
                                                     float* __restrict__ ctxp, float* __restrict__ sump, int N,
                                                     int slab) {
  constexpr int VEC = Elem<T>::kVec;
  constexpr int NV = 32 / VEC;               // vectors per (pixel, head): bf16 4, f32 8
  constexpr int P = 256 / NV;                // pixels staged per iteration: 64 / 32
  __shared__ __attribute__((aligned(16))) float ek[64][32];
  __shared__ __attribute__((aligned(16))) float vv[64][32];
  const int sp = blockIdx.x, h = blockIdx.y, b = blockIdx.z, ns = gridDim.x;
  const int tid = threadIdx.x, d = tid >> 3, e0 = (tid & 7) * 4;
  const int p0 = sp * slab, p1 = min(N, p0 + slab);
  const int sv = tid % NV, spx = tid / NV;   // staging role
  float km[VEC];
#pragma unroll
  for (int u = 0; u < VEC; ++u) km[u] = kmax[(size_t)b * 128 + h * 32 + sv * VEC + u];
  using A = typename Acc<T>::type;            // parity mode: float64 sums over the slab's pixels
  A acc[4] = {0, 0, 0, 0}, ssum = 0;
  const T* base = qkv + (size_t)b * N * 384 + h * 32 + sv * VEC;
  for (int t0 = p0; t0 < p1; t0 += P) {
    const int n = t0 + spx;
    Vec16<T> kvec = vec_zero<T>(), vvec = vec_zero<T>();
    const bool ok = n < p1;
    if (ok) {
      kvec = vec_load(base + (size_t)n * 384 + 128);
      vvec = vec_load(base + (size_t)n * 384 + 256);
    }
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      ek[spx][sv * VEC + u] = ok ? expf(Elem<T>::load(kvec.e[u]) - km[u]) : 0.0f;
      vv[spx][sv * VEC + u] = Elem<T>::load(vvec.e[u]);
    }
    __syncthreads();
#pragma unroll 8
    for (int pp = 0; pp < P; ++pp) {
      const float a = ek[pp][d];
      const float4 w = *reinterpret_cast<const float4*>(&vv[pp][e0]);
      ssum += (A)a;
      acc[0] += (A)a * (A)w.x;
      acc[1] += (A)a * (A)w.y;
      acc[2] += (A)a * (A)w.z;
      acc[3] += (A)a * (A)w.w;
    }
    __syncthreads();
  }
  float* o = ctxp + (((size_t)b * 4 + h) * ns + sp) * 1024 + d * 32 + e0;
  o[0] = (float)acc[0]; o[1] = (float)acc[1]; o[2] = (float)acc[2]; o[3] = (float)acc[3];
  if ((tid & 7) == 0) sump[(((size_t)b * 4 + h) * ns + sp) * 32 + d] = (float)ssum;
}

// grid (4, B), 256 threads x 4 entries: ctx = (sum_split ctxp) / (sum_split sump[d]) / N
__global__ __launch_bounds__(256) void la_fin_kernel(const float* __restrict__ ctxp, const float* __restrict__ sump,
                                                     float* __restrict__ ctx, int ns, int N) {
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const size_t bh = (size_t)b * 4 + h;
  const float invN = 1.0f / (float)N;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + i * 256, d = idx >> 5;
    double s = 0.0, a = 0.0;
    for (int k = 0; k < ns; ++k) {
      a += (double)ctxp[(bh * ns + k) * 1024 + idx];
      s += (double)sump[(bh * ns + k) * 32 + d];
    }
    ctx[bh * 1024 + idx] = (float)a / (float)s * invN;
  }
}

// grid (chunks, B): thread = (pixel, head); ctx of the image's 4 heads in LDS (head pitch 1028 floats: 16-byte aligned
// and the four heads a wave touches fall on different bank quads)
template <typename T>
__global__ __launch_bounds__(256) void la_out_kernel(const T* __restrict__ qkv, const float* __restrict__ ctx,
                                                     T* __restrict__ out, int N) {
  constexpr int VEC = Elem<T>::kVec;
  __shared__ __attribute__((aligned(16))) float cs[4 * 1028];
  const int b = blockIdx.y, tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 256) cs[(i >> 10) * 1028 + (i & 1023)] = ctx[(size_t)b * 4096 + i];
  __syncthreads();
  const int h = tid & 3;
  const float* c = cs + h * 1028;
  for (int n = blockIdx.x * 64 + (tid >> 2); n < N; n += gridDim.x * 64) {
    const T* qp = qkv + ((size_t)b * N + n) * 384 + h * 32;
    float q[32];
#pragma unroll
    for (int k = 0; k < 32 / VEC; ++k) {
      Vec16<T> t = vec_load(qp + k * VEC);
#pragma unroll
      for (int u = 0; u < VEC; ++u) q[k * VEC + u] = Elem<T>::load(t.e[u]);
    }
    float m = q[0];
#pragma unroll
    for (int k = 1; k < 32; ++k) m = fmaxf(m, q[k]);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 32; ++k) { q[k] = expf(q[k] - m); s += q[k]; }
    const float inv = 1.0f / s;
#pragma unroll
    for (int k = 0; k < 32; ++k) q[k] = q[k] * inv * 0.17677669529663687f;  // softmax, then * 32^-0.5
    float o[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) o[e] = 0.0f;
#pragma unroll 2
    for (int dd = 0; dd < 32; ++dd) {
      const float qd = q[dd];
#pragma unroll
      for (int e4 = 0; e4 < 8; ++e4) {
        const float4 cv = *reinterpret_cast<const float4*>(c + dd * 32 + e4 * 4);
        o[e4 * 4 + 0] = fmaf(cv.x, qd, o[e4 * 4 + 0]);
        o[e4 * 4 + 1] = fmaf(cv.y, qd, o[e4 * 4 + 1]);
        o[e4 * 4 + 2] = fmaf(cv.z, qd, o[e4 * 4 + 2]);
        o[e4 * 4 + 3] = fmaf(cv.w, qd, o[e4 * 4 + 3]);
      }
    }
    T* op = out + ((size_t)b * N + n) * 128 + h * 32;
#pragma unroll
    for (int k = 0; k < 32 / VEC; ++k) {
      Vec16<T> w;
#pragma unroll
      for (int u = 0; u < VEC; ++u) w.e[u] = Elem<T>::store(o[k * VEC + u]);
      vec_store(op + k * VEC, w);
    }
  }
}

template <typename T>
int launch_linear_attention(const T* qkv, T* out, float* ws, int B, int N, hipStream_t s) {
  PRG_CHECK(qkv && out && ws && B > 0 && N > 0, "linear attention: bad arguments");
  const int ns = la_nsplit(N);
  const int slab = ceil_div(N, ns);
  float* kmax = ws;
  float* ctxp = kmax + (size_t)B * 128;
  float* sump = ctxp + (size_t)B * 4 * ns * 1024;
  float* ctx = sump + (size_t)B * 4 * ns * 32;
  fill_u32_kernel<<<ceil_div(B * 128, 256), 256, 0, s>>>(reinterpret_cast<uint32_t*>(kmax), 0xFF800000u,
                                                        (size_t)B * 128);
  PRG_LAUNCH_CHECK();
  {
    const int kslab = 128, kns = ceil_div(N, kslab);   // fine slabs: the merge is an atomic max, no reduce pass
    la_kmax_kernel<T><<<dim3(kns, B), 256, 0, s>>>(qkv, kmax, N, kslab);
  }
  PRG_LAUNCH_CHECK();
  la_ctx_kernel<T><<<dim3(ns, 4, B), 256, 0, s>>>(qkv, kmax, ctxp, sump, N, slab);
  PRG_LAUNCH_CHECK();
  la_fin_kernel<<<dim3(4, B), 256, 0, s>>>(ctxp, sump, ctx, ns, N);
  PRG_LAUNCH_CHECK();
  int chunks = ceil_div(N, 64);
  if (chunks > 256) chunks = 256;
  la_out_kernel<T><<<dim3(chunks, B), 256, 0, s>>>(qkv, ctx, out, N);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}
template int launch_linear_attention<float>(const float*, float*, float*, int, int, hipStream_t);
template int launch_linear_attention<bf16_t>(const bf16_t*, bf16_t*, float*, int, int, hipStream_t);

// =============================================================================================
// bottleneck attention core (sd:789-795): one thread per query, keys/values of a (b, head) staged through LDS in
// chunks of 256; two passes (row max, then exp / sum / PV) = the reference's softmax order.
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void full_attn_kernel(const T* __restrict__ qkv, T* __restrict__ out, int N) {
  constexpr int VEC = Elem<T>::kVec;
  constexpr int KC = 128;
  __shared__ __attribute__((aligned(16))) float Ks[KC][32];
  __shared__ __attribute__((aligned(16))) float Vs[KC][32];
  const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int i = blockIdx.x * 256 + tid;
  const int iq = i < N ? i : N - 1;
  const T* base = qkv + (size_t)b * N * 384;
  float q[32];
  {
    const T* qp = base + (size_t)iq * 384 + h * 32;
#pragma unroll
    for (int k = 0; k < 32 / VEC; ++k) {
      Vec16<T> t = vec_load(qp + k * VEC);
#pragma unroll
      for (int u = 0; u < VEC; ++u) q[k * VEC + u] = Elem<T>::load(t.e[u]) * 0.17677669529663687f;
    }
  }
  using A = typename Acc<T>::type;            // parity mode: float64 sums over the keys
  float m = -INFINITY;
  A l = 0, acc[32];
#pragma unroll
  for (int e = 0; e < 32; ++e) acc[e] = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int j0 = 0; j0 < N; j0 += KC) {
      const int cnt = min(KC, N - j0);
      __syncthreads();
      for (int idx = tid; idx < KC * 32; idx += 256) {
        const int j = idx >> 5, ch = idx & 31;
        float kvv = 0.0f, vvv = 0.0f;
        if (j < cnt) {
          kvv = Elem<T>::load(base[(size_t)(j0 + j) * 384 + 128 + h * 32 + ch]);
          if (pass) vvv = Elem<T>::load(base[(size_t)(j0 + j) * 384 + 256 + h * 32 + ch]);
        }
        Ks[j][ch] = kvv;
        if (pass) Vs[j][ch] = vvv;
      }
      __syncthreads();
      for (int j = 0; j < cnt; ++j) {
        float sdot = 0.0f;
#pragma unroll
        for (int k = 0; k < 32; ++k) sdot = fmaf(q[k], Ks[j][k], sdot);
        if (pass == 0) {
          m = fmaxf(m, sdot);
        } else {
          const float pj = expf(sdot - m);
          l += (A)pj;
#pragma unroll
          for (int e = 0; e < 32; ++e) acc[e] += (A)pj * (A)Vs[j][e];
        }
      }
    }
  }
  if (i < N) {
    const float lf = (float)l;
    T* op = out + ((size_t)b * N + i) * 128 + h * 32;
#pragma unroll
    for (int k = 0; k < 32 / VEC; ++k) {
      Vec16<T> w;
#pragma unroll
      for (int u = 0; u < VEC; ++u) w.e[u] = Elem<T>::store((float)acc[k * VEC + u] / lf);
      vec_store(op + k * VEC, w);
    }
  }
}

template <typename T>
int launch_full_attention(const T* qkv, T* out, int B, int N, hipStream_t s) {
  PRG_CHECK(qkv && out && B > 0 && N > 0, "attention: bad arguments");
  full_attn_kernel<T><<<dim3(ceil_div(N, 256), 4, B), 256, 0, s>>>(qkv, out, N);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}
template int launch_full_attention<float>(const float*, float*, int, int, hipStream_t);
template int launch_full_attention<bf16_t>(const bf16_t*, bf16_t*, int, int, hipStream_t);

// =============================================================================================
// conditioning MLP pieces (float32, tiny)
// =============================================================================================
__device__ inline float act_apply(float v, int act) {
  if (act == ACT_SILU) return silu_f(v);
  if (act == ACT_GELU) return gelu_erf_f(v);
  return v;
}

// y[r][o] = act_out(sum_i act_in(x[r][i]) W[o][i] + bias[o]).  Lane = row r (64 rows per block), wave = 4 outputs, the sum
// runs over i IN ORDER in float64 (one rounding at the end: every (r, o) is the same serial sum whatever the launch
// shape).  x goes through LDS transposed ([i][r]: the activation is applied once per element and block, every lane reads
// its own bank), W[o][i] is wave-uniform: scalar loads.  (Round 1-2's one-thread-per-output kernel read W with a stride
// of one row per lane and re-applied the activation O times: 65-86 us per call against ~5.)
constexpr int kLinRows = 64, kLinChunk = 256, kLinOutPerWave = 2;   // (one LDS fill for I <= 256: 66.5 KB)
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, int ldx, int xoff, const float* __restrict__ W, int ldw,
                                                     int woff, const float* __restrict__ bias, float* __restrict__ y, int ldy, int R, int I,
                                                     int O, int act_in, int act_out) {
  extern __shared__ float lin_smem[];
  float (*xs)[kLinRows + 1] = reinterpret_cast<float (*)[kLinRows + 1]>(lin_smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r0 = blockIdx.y * kLinRows, r = r0 + lane;
  const int o0 = (blockIdx.x * 4 + wave) * kLinOutPerWave;
  double acc[kLinOutPerWave];
#pragma unroll
  for (int j = 0; j < kLinOutPerWave; ++j) acc[j] = 0.0;
  for (int i0 = 0; i0 < I; i0 += kLinChunk) {
    __syncthreads();
    // 64 rows x up to 256 columns: thread t loads columns (t & 63) + 64 c of rows t >> 6, + 4, ... (coalesced along i)
    const int n = min(kLinChunk, I - i0);
    for (int c0 = 0; c0 < n; c0 += 64)
#pragma unroll 4
      for (int rr = threadIdx.x >> 6; rr < kLinRows; rr += 4) {
        const int ic = c0 + lane;
        float v = 0.0f;
        if (r0 + rr < R && ic < n) v = act_apply(x[(size_t)(r0 + rr) * ldx + xoff + i0 + ic], act_in);
        if (ic < kLinChunk) xs[ic][rr] = v;
      }
    __syncthreads();
    const float* wr[kLinOutPerWave];
#pragma unroll
    for (int j = 0; j < kLinOutPerWave; ++j)             // (wave-uniform; outputs past O recompute the last one and are not stored)
      wr[j] = W + (size_t)min(o0 + j, O - 1) * ldw + woff + i0;
    auto step = [&](const int i) {
      const double xv = (double)xs[i][lane];
#pragma unroll
      for (int j = 0; j < kLinOutPerWave; ++j) acc[j] += xv * (double)wr[j][i];
    };
    int i = 0;
    for (; i + 16 <= n; i += 16)
#pragma unroll
      for (int u = 0; u < 16; ++u) step(i + u);          // (unrolled: the scalar loads of 16 steps are in flight together)
    for (; i < n; ++i) step(i);
  }
  if (r < R) {
#pragma unroll
    for (int j = 0; j < kLinOutPerWave; ++j) {
      const int o = o0 + j;
      if (o < O) {
        float a = (float)acc[j];
        if (bias) a += bias[o];
        y[(size_t)r * ldy + o] = act_apply(a, act_out);
      }
    }
  }
}

constexpr size_t kLinSmem = sizeof(float) * kLinChunk * (kLinRows + 1);

int launch_linear(const float* x, int ldx, int xoff, const float* W, int ldw, int woff, const float* bias, float* y,
                  int ldy, int R, int I, int O, int act_in, int act_out, hipStream_t s) {
  PRG_CHECK(x && W && y && R > 0 && I > 0 && O > 0, "linear: bad arguments");
  const dim3 grid(ceil_div(O, 4 * kLinOutPerWave), ceil_div(R, kLinRows));
  static DeviceOnce attr;   // one-time opt-in; atomic: lanes launch from several host threads (idempotent call)
  if (!attr.done()) {
    PRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLinSmem));
    attr.mark();
  }
  linear_kernel<<<grid, 256, kLinSmem, s>>>(x, ldx, xoff, W, ldw, woff, bias, y, ldy, R, I, O, act_in, act_out);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

// freqs[i] = exp(-i ln(1e4) / (half - 1)) comes from the HOST (prg_unet_set_time_freqs): the reference evaluates that
// float32 exp with torch on whatever device it runs on, and a 1-ulp difference in a frequency, multiplied by t <= 999,
// moves every time embedding by up to 6e-5 — enough to shift a 50-step chain by 4e-5 (measured: the same reference code on
// two CPUs).  Taking the table as data keeps this library on the caller's side of that dependence.
template <typename TI>
__global__ void sinusoidal_kernel(const TI* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ out, int R,
                                  int dim) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (idx >= R * half) return;
  const int r = idx / half, i = idx - r * half;
  const float a = (float)t[r] * freqs[i];
  out[(size_t)r * dim + i] = sinf(a);
  out[(size_t)r * dim + half + i] = cosf(a);
}

int launch_sinusoidal(const int64_t* t, const float* freqs, float* out, int R, int dim, hipStream_t s) {
  PRG_CHECK(t && freqs && out && R > 0 && dim >= 4 && dim % 2 == 0, "sinusoidal: bad arguments");
  sinusoidal_kernel<int64_t><<<ceil_div(R * (dim / 2), 256), 256, 0, s>>>(t, freqs, out, R, dim);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}
int launch_sinusoidal_i32(const int32_t* t, const float* freqs, float* out, int R, int dim, hipStream_t s) {
  PRG_CHECK(t && freqs && out && R > 0 && dim >= 4 && dim % 2 == 0, "sinusoidal: bad arguments");
  sinusoidal_kernel<int32_t><<<ceil_div(R * (dim / 2), 256), 256, 0, s>>>(t, freqs, out, R, dim);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}

// =============================================================================================
// debug taps
// =============================================================================================
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ out, int HW, int C, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / ((size_t)HW * C), rem = i - b * HW * C;
    const int c = (int)(rem / HW), p = (int)(rem - (size_t)c * HW);
    out[i] = Elem<T>::load(x[(b * HW + p) * C + c]);
  }
}
template <typename T>
int launch_nhwc_to_nchw_f32(const T* x, float* out, int B, int HW, int C, hipStream_t s) {
  const size_t n = (size_t)B * HW * C;
  int grid = (int)((n + 255) / 256);
  if (grid > 4096) grid = 4096;
  nhwc_to_nchw_kernel<T><<<grid, 256, 0, s>>>(x, out, HW, C, n);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ out, int HW, int C, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / ((size_t)HW * C), rem = i - b * HW * C;
    const int p = (int)(rem / C), c = (int)(rem - (size_t)p * C);
    out[i] = Elem<T>::store(x[(b * C + c) * HW + p]);
  }
}
template <typename T>
int launch_nchw_f32_to_nhwc(const float* x, T* out, int B, int HW, int C, hipStream_t s) {
  const size_t n = (size_t)B * HW * C;
  int grid = (int)((n + 255) / 256);
  if (grid > 4096) grid = 4096;
  nchw_to_nhwc_kernel<T><<<grid, 256, 0, s>>>(x, out, HW, C, n);
  PRG_LAUNCH_CHECK();
  return PRG_OK;
}
template int launch_nchw_f32_to_nhwc<float>(const float*, float*, int, int, int, hipStream_t);
template int launch_nchw_f32_to_nhwc<bf16_t>(const float*, bf16_t*, int, int, int, hipStream_t);
template int launch_nhwc_to_nchw_f32<float>(const float*, float*, int, int, int, hipStream_t);
template int launch_nhwc_to_nchw_f32<bf16_t>(const bf16_t*, float*, int, int, int, hipStream_t);

}  // namespace prg
