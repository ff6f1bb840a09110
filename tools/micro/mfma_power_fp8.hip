// Power-limited ceiling of v_mfma_scale_f32_32x32x64_f8f6f4 (MX e4m3 operands) with random data, next to the bf16 MFMA
// (mfma_power.hip): does the fp8 instruction deliver its 2x on this part's power budget?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ inline unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

template <int WPS>
__global__ __launch_bounds__(256 * WPS) void k(float* out, long long* clk, long long* wall, int iters, int random) {
  unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 97u + 1u;
  i32x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      // random e4m3 bytes with exponent fields kept mid-range (no NaN: 0x7f / 0xff excluded by masking the top exponent bit)
      a[i][j] = random ? (int)(lcg(seed) & 0xB7B7B7B7u) : 0x38383838;
      b[i][j] = random ? (int)(lcg(seed) & 0xB7B7B7B7u) : 0x30303030;
    }
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
  long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[i], acc[i], 0, 0, 0, 127, 0, 120);
  }
  long long t1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int q = 0; q < 4; ++q) for (int e = 0; e < 16; ++e) s += acc[q][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; }
}

template <int WPS>
void run(int random) {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int grid = p.multiProcessorCount, iters = 40000;
  float* out; long long *clk, *wall;
  (void)hipMalloc(&out, grid * 256 * WPS * sizeof(float)); (void)hipMalloc(&clk, grid * 8); (void)hipMalloc(&wall, grid * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<WPS><<<grid, 256 * WPS>>>(out, clk, wall, iters, random);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c, w; (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&w, wall, 8, hipMemcpyDeviceToHost);
    const double flops = (double)grid * 4 * WPS * iters * 4 * 2.0 * 32 * 32 * 64;
    printf("fp8 waves/SIMD %d  %s operands: %.2f ms  %.0f TFLOP/s  shader clock %.0f MHz  %.1f clk per MFMA per wave\n", WPS,
           random ? "random  " : "constant", ms, flops / ms / 1e9, (double)c / ((double)w / 100.0), (double)c / (iters * 4.0));
  }
}
int main() { run<1>(0); run<1>(1); run<2>(0); run<2>(1); return 0; }
