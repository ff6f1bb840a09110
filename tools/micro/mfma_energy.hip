// What does an MFMA cost on a power-capped MI355X?  Register-only loops (no LDS, no memory), one or two waves per SIMD on every CU, ~1 s per
// configuration so that the power management settles; reports sustained TFLOP/s and the shader clock (clock64 / wall_clock64).
//   shape     32x32x16 bf16 | 32x32x16 f16 | 16x16x32 bf16
//   operands  random (8 distinct A and 8 distinct B fragments per wave, cycled) | constant
//   order     distinct : MFMA i multiplies (A[i], B[i])                     — every MFMA changes both operands
//             shareA4  : four consecutive MFMAs keep A, change B           — conv3x3_c64_kernel's round-5 consumer loop
//             shareB3  : three consecutive MFMAs keep B, change A          — its row-reuse loop
//             pairs    : two consecutive MFMAs multiply the SAME (A, B)    — the PRG_C64_EXP=2048 timing experiment
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_energy tools/micro/mfma_energy.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ inline unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

enum { DISTINCT = 0, SHARE_A4 = 1, SHARE_B3 = 2, PAIRS = 3 };

// SHAPE 0: 32x32x16 bf16, 1: 32x32x16 f16, 2: 16x16x32 bf16 (two of them per slot: the same FLOPs per slot in every shape)
template <int WPS, int SHAPE, int ORDER>
__global__ __launch_bounds__(256 * WPS) void k(float* out, long long* clk, long long* wall, int iters, int random) {
  unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 97u + 1u;
  bf16x8 a[8], b[8];
  f16x8 ah[8], bh[8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) {
      const float va = random ? ((lcg(seed) >> 8) * (1.0f / 8388608.0f) - 1.0f) : 0.5f;
      const float vb = random ? ((lcg(seed) >> 8) * (1.0f / 8388608.0f) - 1.0f) : 0.25f;
      a[i][j] = (__bf16)va; b[i][j] = (__bf16)vb;
      ah[i][j] = (_Float16)va; bh[i][j] = (_Float16)vb;
    }
  f32x16 acc[4];
  f32x4 acc4[8];
  for (int q = 0; q < 4; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
  for (int q = 0; q < 8; ++q) for (int e = 0; e < 4; ++e) acc4[q][e] = 0.f;
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 24; ++i) {                       // 24 slots: a multiple of 2, 3, 4 and 8
      const int q = i & 3;
      const int ia = ORDER == SHARE_A4 ? (i >> 2) & 7 : ORDER == PAIRS ? (i >> 1) & 7 : i & 7;
      const int ib = ORDER == SHARE_B3 ? (i / 3) & 7 : ORDER == PAIRS ? (i >> 1) & 7 : ORDER == SHARE_A4 ? (i * 3 + 1) & 7 : (i * 5 + 3) & 7;
      if constexpr (SHAPE == 0) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia], b[ib], acc[q], 0, 0, 0);
      if constexpr (SHAPE == 1) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ia], bh[ib], acc[q], 0, 0, 0);
      if constexpr (SHAPE == 2) {
        acc4[2 * q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ia], b[ib], acc4[2 * q], 0, 0, 0);
        acc4[2 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ia], b[(ib + 1) & 7], acc4[2 * q + 1], 0, 0, 0);
      }
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int q = 0; q < 4; ++q) for (int e = 0; e < 16; ++e) s += acc[q][e];
  for (int q = 0; q < 8; ++q) for (int e = 0; e < 4; ++e) s += acc4[q][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; }
}

template <int WPS, int SHAPE, int ORDER>
void run(int random, const char* what) {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int grid = p.multiProcessorCount;
  float* out; long long *clk, *wall;
  (void)hipMalloc(&out, grid * 256 * WPS * sizeof(float)); (void)hipMalloc(&clk, grid * 8); (void)hipMalloc(&wall, grid * 8);
  const int iters = 50000 / WPS;                         // ~40-60 ms per launch
  const double flops_launch = (double)grid * 4 * WPS * iters * 24 * 2.0 * 32 * 32 * 16;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 6; ++r) k<WPS, SHAPE, ORDER><<<grid, 256 * WPS>>>(out, clk, wall, iters, random);   // settle (~0.3 s)
  (void)hipDeviceSynchronize();
  const int reps = 12;
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) k<WPS, SHAPE, ORDER><<<grid, 256 * WPS>>>(out, clk, wall, iters, random);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c, w; (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&w, wall, 8, hipMemcpyDeviceToHost);
  printf("%-46s waves/SIMD %d  %-8s  %7.0f TFLOP/s  clock %4.0f MHz  %.1f ms\n", what, WPS, random ? "random" : "constant",
         flops_launch * reps / ms / 1e9, (double)c / ((double)w / 100.0), ms);
  (void)hipFree(out); (void)hipFree(clk); (void)hipFree(wall);
}

int main() {
  for (int round = 0; round < 2; ++round) {
    printf("-- round %d\n", round + 1);
    run<1, 0, DISTINCT>(0, "32x32x16 bf16  distinct");
    run<1, 0, DISTINCT>(1, "32x32x16 bf16  distinct");
    run<1, 0, SHARE_A4>(1, "32x32x16 bf16  A kept for 4 MFMAs");
    run<1, 0, SHARE_B3>(1, "32x32x16 bf16  B kept for 3 MFMAs");
    run<1, 0, PAIRS>(1, "32x32x16 bf16  pairs of identical MFMAs");
    run<1, 1, DISTINCT>(1, "32x32x16 f16   distinct");
    run<1, 1, SHARE_A4>(1, "32x32x16 f16   A kept for 4 MFMAs");
    run<1, 2, DISTINCT>(1, "16x16x32 bf16  distinct (two per slot)");
    run<1, 2, SHARE_A4>(1, "16x16x32 bf16  A kept for 4 slots");
    run<2, 0, DISTINCT>(1, "32x32x16 bf16  distinct");
    run<2, 0, SHARE_A4>(1, "32x32x16 bf16  A kept for 4 MFMAs");
  }
  return 0;
}
