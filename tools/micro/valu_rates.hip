// Calibration: issue cost of VALU instruction kinds on gfx950, one wave per SIMD (256 threads per CU), 16 independent
// accumulator chains per lane: v_fmac_f32, v_pk_fma_f32, v_dot2c_f32_bf16, v_exp_f32, v_cvt_pk_bf16_f32, v_add_f32 DPP.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters) {
  float a[16];
  f2 p[8];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 8; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
  const float x = 1.0001f + threadIdx.x * 1e-7f, y = 0.5f;
  const unsigned bx = __builtin_bit_cast(unsigned, bf2{(__bf16)x, (__bf16)y});
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
    } else if (KIND == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(f2{x, x}), "v"(f2{y, y}));
    } else if (KIND == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(bx), "v"(bx));
    } else if (KIND == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
    } else if (KIND == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
    } else if (KIND == 5) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
    } else if (KIND == 6) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(f2{x, y}));
    } else if (KIND == 7) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
    } else if (KIND == 8) {       // round 4 (h16): the f16 conversions and packed / transcendental f16 instructions
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
    } else if (KIND == 9) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
    } else if (KIND == 10) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(bx), "v"(bx));
    } else if (KIND == 11) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
    } else if (KIND == 12) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(a[i]));
    } else if (KIND == 13) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(a[i]));
    }
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char* name, int n_instr, float* out, long long* clk) {
  const int iters = 20000;
  k<KIND><<<256, 256>>>(out, clk, iters);
  (void)hipDeviceSynchronize();
  long long c;
  (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  printf("%-22s %.2f clk per instruction (one wave per SIMD)\n", name, (double)c / iters / n_instr);
}
int main() {
  float* out; long long* clk;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&clk, 256 * 8);
  run<0>("v_fmac_f32", 16, out, clk);
  run<1>("v_pk_fma_f32", 8, out, clk);
  run<6>("v_pk_add_f32", 8, out, clk);
  run<2>("v_dot2c_f32_bf16", 16, out, clk);
  run<3>("v_exp_f32", 16, out, clk);
  run<7>("v_rcp_f32", 16, out, clk);
  run<4>("v_cvt_pk_bf16_f32", 16, out, clk);
  run<5>("v_add_f32 dpp quad", 16, out, clk);
  run<8>("v_cvt_pk_f16_f32", 16, out, clk);
  run<9>("v_cvt_pkrtz_f16_f32", 16, out, clk);
  run<13>("v_cvt_f16_f32", 16, out, clk);
  run<10>("v_pk_fma_f16", 16, out, clk);
  run<11>("v_exp_f16", 16, out, clk);
  run<12>("v_rcp_f16 sdwa hi", 16, out, clk);
  return 0;
}
