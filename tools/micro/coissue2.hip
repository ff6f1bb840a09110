// Co-issue calibration, second take: the VALU wave is THROUGHPUT-bound (32 independent chains, like the 88 independent
// elements of a halo transform), and the MFMA accumulators live either in arch VGPRs (what hipcc picks below 256
// registers) or in AccVGPRs (inline asm, "a" constraint).  block = 512 threads: waves 0-3 MFMA, waves 4-7 VALU.
// mode bit0: MFMA waves active, bit1: VALU waves active, bit2: no transcendentals, bit3: accumulators in AGPRs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <bool AG>
__device__ __forceinline__ void mm(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

template <bool AG>
__global__ __launch_bounds__(512) void k(float* out, long long* clk, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  long long t0 = clock64();
  float s = 0;
  if (wave < 4) {
    if (mode & 1) {
      bf16x8 a, b;
      for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
      f32x16 acc[4];
      for (int q = 0; q < 4; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) mm<AG>(acc[q], a, b);
      }
      for (int q = 0; q < 4; ++q) for (int e = 0; e < 16; ++e) s += acc[q][e];
    }
  } else if (mode & 2) {
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = threadIdx.x * 0.01f + i;
    for (int it = 0; it < iters / 2; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        float y = fmaf(x[i], 1.0001f, 0.001f);
        if (mode & 4) x[i] = fmaf(y, 0.999f, y * 0.0001f) + 0.5f * y;
        else x[i] = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.44f));
      }
    }
    for (int i = 0; i < 32; ++i) s += x[i];
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  int grid = p.multiProcessorCount, iters = 10000;
  float* out; long long* clk;
  (void)hipMalloc(&out, grid * 512 * sizeof(float)); (void)hipMalloc(&clk, grid * 8 * sizeof(long long));
  for (int mode : {1, 2, 3, 6, 7, 9, 11, 15, 1}) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    if (mode & 8) k<true><<<grid, 512>>>(out, clk, iters, mode);
    else k<false><<<grid, 512>>>(out, clk, iters, mode);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c[8]; (void)hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
    printf("mode %2d: %.3f ms   MFMA wave: %.1f clk/MFMA   VALU wave: %.1f clk per element\n", mode, ms,
           (double)c[0] / (iters * 4.0), (double)c[4] / (iters / 2 * 32.0));
  }
  return 0;
}
