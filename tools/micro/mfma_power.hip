// Power-limited MFMA ceiling with RANDOM operand data (mfma_peak.hip uses near-constant operands, which toggle few bits).
// Each wave cycles through 8 operand pairs of random bf16 values; accumulators stay bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ inline unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

template <int WPS>
__global__ __launch_bounds__(256 * WPS) void k(float* out, long long* clk, long long* wall, int iters, int random) {
  unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 97u + 1u;
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) {
      float va = random ? ((lcg(seed) >> 8) * (1.0f / 8388608.0f) - 1.0f) : 0.5f;
      float vb = random ? ((lcg(seed) >> 8) * (1.0f / 8388608.0f) - 1.0f) : 0.25f;
      a[i][j] = (__bf16)va; b[i][j] = (__bf16)vb;
    }
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
  long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = i & 3;
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], acc[q], 0, 0, 0);
    }
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
    const double flops = (double)grid * 4 * WPS * iters * 8 * 2.0 * 32 * 32 * 16;
    printf("waves/SIMD %d  %s operands: %.2f ms  %.0f TFLOP/s  shader clock %.0f MHz  %.1f clk per MFMA per wave\n", WPS,
           random ? "random  " : "constant", ms, flops / ms / 1e9, (double)c / ((double)w / 100.0), (double)c / (iters * 8.0));
  }
}
int main() { run<1>(0); run<1>(1); run<2>(0); run<2>(1); return 0; }
