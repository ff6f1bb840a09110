// Calibration microbenchmark: achievable v_mfma_f32_32x32x16_bf16 rate on this device and the tick rate of clock64().
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, long long* clk, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  f32x16 acc[NACC];
  for (int k = 0; k < NACC; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0;
  for (int k = 0; k < NACC; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  int blocks_per_cu = argc > 1 ? atoi(argv[1]) : 1;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int cus = p.multiProcessorCount;
  int grid = cus * blocks_per_cu, iters = 20000;
  float* out; long long* clk;
  hipMalloc(&out, grid * 256 * sizeof(float)); hipMalloc(&clk, grid * sizeof(long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    mfma_loop<4><<<grid, 256>>>(out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, clk, sizeof(c), hipMemcpyDeviceToHost);
    double mfmas = (double)grid * 4 /*waves*/ * iters * 4;
    double flops = mfmas * 2.0 * 32 * 32 * 16;
    printf("cus=%d blocks/cu=%d clockRate=%d kHz: %.3f ms  %.1f TFLOP/s  clock64 ticks per MFMA (per wave) = %.2f  ticks/us = %.1f\n",
           cus, blocks_per_cu, p.clockRate, ms, flops / ms / 1e9, (double)c / (iters * 4.0), (double)c / (ms * 1e3));
  }
  return 0;
}
