// Co-issue calibration: does VALU work of one wave overlap MFMA work of another wave on the same SIMD?
// block = 512 threads (8 waves, two per SIMD): waves 0-3 run MFMAs, waves 4-7 run a VALU (fma + exp2 + rcp) loop.
// mode bit0: MFMA waves active, bit1: VALU waves active, bit2: VALU loop without transcendentals.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

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
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
      }
      for (int q = 0; q < 4; ++q) for (int e = 0; e < 16; ++e) s += acc[q][e];
    }
  } else if (mode & 2) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.01f + i;
    // per iteration: 8 channels of the fused prologue (fma, exp2, add, rcp, mul) ~ what a halo unit costs
    for (int it = 0; it < iters * 2; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float y = fmaf(x[i], 1.0001f, 0.001f);
        if (mode & 4) x[i] = fmaf(y, 0.999f, y * 0.0001f) + 0.5f * y;
        else x[i] = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.44f));
      }
    }
    for (int i = 0; i < 8; ++i) s += x[i];
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
  for (int mode : {1, 2, 3, 6, 7, 1}) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<<<grid, 512>>>(out, clk, iters, mode);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c[8]; (void)hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
    printf("mode %d: %.3f ms   MFMA wave: %.1f clk/MFMA   VALU wave: %.1f clk per 8-channel unit\n", mode, ms,
           (double)c[0] / (iters * 4.0), (double)c[4] / (iters * 2.0));
  }
  return 0;
}
