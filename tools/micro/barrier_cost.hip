// Calibration: cost of one s_barrier round for a 512-thread workgroup (8 waves, 2 per SIMD), alone and with a few
// instructions of work per wave between barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(long long* clk, float* out, int iters, int work) {
  float x = threadIdx.x * 0.001f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    for (int w = 0; w < work; ++w) x = fmaf(x, 1.0001f, 0.5f);
    __builtin_amdgcn_s_barrier();
  }
  long long t1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = x;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
  long long* clk; float* out;
  (void)hipMalloc(&clk, 256 * 8); (void)hipMalloc(&out, 256 * 512 * 4);
  for (int work : {0, 8, 32, 128}) {
    k<<<256, 512>>>(clk, out, 10000, work);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("work %3d fma per wave per phase: %.1f clk per barrier round\n", work, (double)c / 10000);
  }
  return 0;
}
