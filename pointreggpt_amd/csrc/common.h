// common.h — shared device/host helpers for libprg_hip (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <string>

#include "../../include/prg.h"

namespace prg {

// ---------------------------------------------------------------------------------------------
// error plumbing (thread-local message behind prg_last_error)
// ---------------------------------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define PRG_HIP(expr)                                                                            \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess)                                                                        \
      return ::prg::fail(PRG_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));          \
  } while (0)

#define PRG_CHECK(cond, msg)                                                                     \
  do {                                                                                           \
    if (!(cond)) return ::prg::fail(PRG_E_INVALID, std::string(msg) + " [" #cond "]");           \
  } while (0)

#define PRG_LAUNCH_CHECK()                                                                       \
  do {                                                                                           \
    hipError_t _e = hipGetLastError();                                                           \
    if (_e != hipSuccess) return ::prg::fail(PRG_E_HIP, std::string("launch: ") + hipGetErrorString(_e)); \
  } while (0)

// ---------------------------------------------------------------------------------------------
// per-DEVICE one-time state: hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count belong to a device, not to the
// process — a process that drives a second GPU must opt in there too.  One bit per device ordinal; atomic because
// lanes launch from several host threads (the guarded calls are idempotent, a race is benign).
// ---------------------------------------------------------------------------------------------
struct DeviceOnce {
  std::atomic<uint64_t> mask{0};
  static uint64_t bit() {
    int d = 0;
    (void)hipGetDevice(&d);
    return (uint64_t)1 << (d & 63);
  }
  bool done() const { return (mask.load(std::memory_order_acquire) & bit()) != 0; }
  void mark() { mask.fetch_or(bit(), std::memory_order_release); }
};
int device_cu_count();   // multiProcessorCount of the CURRENT device (cached per device ordinal); 0 on failure

// ---------------------------------------------------------------------------------------------
// bf16 storage type (raw 16 bits; conversions round-to-nearest-even, NaN preserved)
// ---------------------------------------------------------------------------------------------
struct bf16_t {
  uint16_t v;
};

__host__ __device__ inline float bf16_to_f32(bf16_t b) {
  union {
    uint32_t u;
    float f;
  } x;
  x.u = ((uint32_t)b.v) << 16;
  return x.f;
}

__host__ __device__ inline bf16_t f32_to_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  // gfx950 has a native round-to-nearest-even conversion (v_cvt_pk_bf16_f32)
  bf16_t r;
  r.v = __builtin_bit_cast(uint16_t, (__bf16)f);
  return r;
#else
  union {
    uint32_t u;
    float f;
  } x;
  x.f = f;
  bf16_t r;
  if ((x.u & 0x7fffffffu) > 0x7f800000u) {  // NaN
    r.v = (uint16_t)((x.u >> 16) | 0x40);
    return r;
  }
  uint32_t lsb = (x.u >> 16) & 1u;
  x.u += 0x7fffu + lsb;
  r.v = (uint16_t)(x.u >> 16);
  return r;
#endif
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int kVec = 4;  // elements per 16-byte vector
  __host__ __device__ static inline float load(float v) { return v; }
  __host__ __device__ static inline float store(float v) { return v; }
  // parity mode: IEEE exp / divide, the reference's x * sigmoid(x)
  __device__ static inline float silu(float x) { return x / (1.0f + expf(-x)); }
};
template <>
struct Elem<bf16_t> {
  static constexpr int kVec = 8;
  __host__ __device__ static inline float load(bf16_t v) { return bf16_to_f32(v); }
  __host__ __device__ static inline bf16_t store(float v) { return f32_to_bf16(v); }
  // throughput mode: v_exp_f32 + v_rcp_f32 (1 ulp-class, far below bf16 resolution)
  __device__ static inline float silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
  }
};

// 16-byte vector of T with element access as float
template <typename T>
struct alignas(16) Vec16 {
  T e[Elem<T>::kVec];
};

template <typename T>
__device__ inline Vec16<T> vec_zero() {
  Vec16<T> v;
  uint4 z = make_uint4(0, 0, 0, 0);
  *reinterpret_cast<uint4*>(&v) = z;
  return v;
}
template <typename T>
__device__ inline Vec16<T> vec_load(const T* p) {
  Vec16<T> v;
  *reinterpret_cast<uint4*>(&v) = *reinterpret_cast<const uint4*>(p);
  return v;
}
template <typename T>
__device__ inline void vec_store(T* p, const Vec16<T>& v) {
  *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(&v);
}

// ---------------------------------------------------------------------------------------------
// wave / block reductions (wave = 64)
// ---------------------------------------------------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

constexpr int kMaxTicketImages = 4096;   // per-image arrival counters (conv kernels folding GroupNorm coefficients): B <= this
constexpr int kGnMaxSplit = 1024;  // max GroupNorm (sum, sumsq) partial slabs per image (256x256: 8x32 tiles x 4 wave rows)

// GroupNorm parameters of one Block (+ the ResnetBlock conditioning): what folds the statistics into y = x * A + B
struct GnApply {
  const float* gamma;     // [C]
  const float* beta;      // [C]
  const float* ss_a;      // conditioning (scale | shift) rows of 2C floats, or null.  value = ss_a[b or step] + ss_b[b]
  const float* ss_b;      // second addend (per image), or null
  int64_t ss_b_stride;    // row stride of ss_b per image
  int64_t ss_a_stride;    // row stride of ss_a per image (0 when shared by the whole batch)
  const int* ss_a_row;    // optional device int: row index into ss_a added on top (sampler: current step), or null
  int64_t ss_a_row_stride;
};

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics as fixed-point accumulators (bf16 throughput path, round 3)
// ---------------------------------------------------------------------------------------------
// Instead of per-tile (sum, sumsq) slabs + one gn_coeff_kernel launch per GroupNorm (38 launches per U-Net evaluation),
// the producing conv adds every wave's partial sums into ONE pair of 64-bit integers per (image, group) with no-return
// agent-scope atomics — integer addition is associative, so the result does not depend on the order in which tiles
// finish: deterministic and batch-slot invariant like the fixed-order slab reduction it replaces — and every CONSUMER
// turns the pair into the affine coefficients itself: y = x * A + B with A = rstd * P[c], B = Q[c] - mean * A, where
// P = gamma * (scale + 1), Q = beta * (scale + 1) + shift were folded once per evaluation (cond_fold_kernel) or at weight
// load (unconditioned norms: P = gamma, Q = beta).  Scales: sum * 2^24, sumsq * 2^20 (|sum| < 5.5e11, sumsq < 8.8e12: an
// image group of 524288 elements with rms < 4096; resolution 6e-8 / 1e-6 per partial).
constexpr float kGnSumScale = 16777216.0f, kGnSqScale = 1048576.0f;

struct GnFold {
  const long long* acc;   // [B][G][2] (sum * 2^24, sumsq * 2^20); null = this launch uses coefficient tables instead
  const float* P;         // [C] (+ b * pq_stride), 16-byte aligned
  const float* Q;
  int64_t pq_stride;      // floats between images (0: shared by the batch)
  float inv_n;            // 1 / (pixels per image * channels per group)
  int G, cpg;             // groups, channels per group (C = G * cpg)
};

#if defined(__HIPCC__)
// one wave total (float) -> the image group's accumulator; `which` 0 = sum, 1 = sumsq
// A NaN / Inf partial must stay visible (float slabs propagated it; __float2ll_rn would turn it into 0 / a saturated value and
// the norm would emit finite, wrong numbers): a non-finite partial sets the SIGN bit of the group's sum-of-squares accumulator
// with an atomic OR — a legitimate sum of squares is non-negative and below 2^61, so concurrent adds never carry into that
// bit and any number of poisoning waves leaves it set — and every consumer's fold answers NaN for such a group.
constexpr unsigned long long kGnPoison = 0x8000000000000000ull;
__device__ inline void gn_acc_add(long long* acc, int G, int b, int g, int which, float v) {
  long long* p = acc + ((size_t)b * G + g) * 2 + which;
  if (!(fabsf(v) <= 3.0e38f)) {                 // NaN or Inf
    __hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(acc + ((size_t)b * G + g) * 2 + 1), kGnPoison, __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const long long q = __float2ll_rn(v * (which ? kGnSqScale : kGnSumScale));
  __hip_atomic_fetch_add(p, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void gn_fold_stats_raw(long long s_fx, long long q_fx, float inv_n, float& mean, float& rstd) {
  const double m = (double)s_fx * (1.0 / 16777216.0) * (double)inv_n;
  double var = (double)q_fx * (1.0 / 1048576.0) * (double)inv_n - m * m;
  var = var < 0.0 ? 0.0 : var;
  mean = (float)m;
  rstd = __builtin_amdgcn_rsqf((float)var + 1e-5f);
  if (q_fx < 0) mean = rstd = __builtin_nanf("");   // poisoned by a non-finite partial (gn_acc_add)
}
__device__ inline void gn_fold_stats(const GnFold& f, int b, int g, float& mean, float& rstd) {
  const long long* a = f.acc + ((size_t)b * f.G + g) * 2;
  gn_fold_stats_raw(a[0], a[1], f.inv_n, mean, rstd);
}
#endif

// ---------------------------------------------------------------------------------------------
// activations
// ---------------------------------------------------------------------------------------------
__device__ inline float silu_f(float x) { return x / (1.0f + expf(-x)); }
__device__ inline float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ inline float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace prg
